// gemm_tn_fp8.hip -- weight-gradient GEMM on fp8 operands for gfx950:  dW[N,K] = dY[M,N]^T . X[M,K]
// (autograd of nn.Linear, vit.py:20,23,44,47; BASELINE config 5 / SURVEY 8f item 2)
//
// dY is OCP e5m2 (the gradient copy the backward already makes for its dX GEMM), X is OCP e4m3 (the activation), both row-major
// with the token index m as the slow one, one byte per element.  The structure is gemm_tn256_kernel's (gemm_bf16.hip): 256 (n) x 256 (k)
// output tile, 8 waves as 2 (n) x 4 (k), wave tile 128 x 64 = 8 x 4 fragments, tiles staged through registers into padded LDS
// rows, split over M into f32 slabs that a second kernel folds deterministically.  What changes with 1-byte elements:
//
//  * fragments come from ds_read_b64_tr_b8: within a 16-lane group, lane i points at 8 bytes -- row i / 2, byte-chunk i % 2 -- of an
//    8 (token rows) x 16 (columns) byte matrix and receives column i, all 8 rows: exactly the 8 consecutive reduction
//    indices one v_mfma_f32_16x16x32 fp8 operand lane holds.  ONE transposing read feeds one MFMA operand (the 16-bit kernel
//    needs two ds_read_b64_tr_b16 per operand, and is bound by their issue rate: 405 cycles of reads per 272 of MFMA), and
//    the tiles are half the bytes.
//  * KB = 32: v_mfma_f32_16x16x32_bf8_fp8 (A = dY e5m2, B = X e4m3), 64 token rows per LDS step.
//    KB = 128: v_mfma_f32_16x16x128_f8f6f4 with unit block scales (twice the matrix rate): a lane's 32 operand bytes are FOUR
//    transposing reads, token rows q * 32 + 8 g .. + 7 (q = 0..3) -- the same assignment in both operands, so the reduction
//    index permutation is a legal one; 128 token rows per LDS step (the LDS footprint of the 16-bit kernel).
//  * rows of 256 bytes + 16 bytes of padding (272): the 16 rows that lanes 0-31 of a transposing read touch start 16 bytes
//    apart modulo 256, i.e. they cover the 64 banks exactly once (the read needs two passes for its 512 bytes anyway).
//  * the per-tensor scales (1 / scale_dy, 1 / scale_x: device scalars of the delayed-scaling state) are applied by the fold.
//
// UNMEASURED when written (no GPU minutes left in round 3): correctness rests on tests/test_fp8_backward_gpu.py.
#include "common.h"
#include <stdlib.h>

namespace {

typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

constexpr int F8T_LD = 272;                 // bytes per LDS row: 256 + 16

__device__ __forceinline__ int f8t_xcd_swizzle(int b, int nwg) {
    const int q = nwg / 8, r = nwg % 8;
    const int xcd = b % 8, idx = b / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ i32x2 tr8(const char* p) {
    return __builtin_amdgcn_ds_read_tr8_b64_v2i32((i32x2 __attribute__((address_space(3)))*)(p));
}

template <int KB>
__global__ __launch_bounds__(512) void gemm_tn256_f8_kernel(const unsigned char* __restrict__ dY, long long ldy,
                                                            const unsigned char* __restrict__ X, long long ldx,
                                                            float* __restrict__ ws, int M, int N, int K, int rows_per_split,
                                                            int tiles_k, int nwg) {
    static_assert(KB == 32 || KB == 128, "tokens per MFMA: 32 or 128");
    constexpr int BKM = KB == 128 ? 128 : 64;           // token rows per LDS step
    constexpr int TILE = BKM * F8T_LD;                  // one operand, one stage: 17,408 / 34,816 B
    constexpr int STAGE = 2 * TILE;
    constexpr int NCH = BKM * 16 / 512;                 // 16-byte chunks per operand per thread and step: 2 / 4
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wn = wave >> 2, wk = wave & 3;            // wave tile: 128 (n) x 64 (k)
    const int lin = f8t_xcd_swizzle(blockIdx.x, (int)gridDim.x);      // (split, tile) jointly, split-major: see gemm_tn256_kernel
    const int split = lin / nwg;
    const int wg = lin % nwg;
    const int tn = wg / tiles_k, tk = wg % tiles_k;
    const int n0 = tn * 256, k0 = tk * 256;
    const int mbeg = split * rows_per_split;
    int mend = mbeg + rows_per_split; mend = mend < M ? mend : M;

    int srow[NCH], scol[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) { const int c = tid + 512 * j; srow[j] = c >> 4; scol[j] = (c & 15) * 16; }
    const i32x4 zero4 = {0, 0, 0, 0};                   // byte 0 is +0 in both formats: rows past the end add nothing
    i32x4 ry[NCH], rx[NCH];
    auto gload = [&](int mb) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int m = mb + srow[j];
            const bool mv = m < mend;
            ry[j] = (mv && n0 + scol[j] < N) ? *reinterpret_cast<const i32x4*>(dY + (long long)m * ldy + n0 + scol[j]) : zero4;
            rx[j] = (mv && k0 + scol[j] < K) ? *reinterpret_cast<const i32x4*>(X + (long long)m * ldx + k0 + scol[j]) : zero4;
        }
    };
    auto lstore = [&](int buf) {
        char* base = lds + buf * STAGE;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            *reinterpret_cast<i32x4*>(base + srow[j] * F8T_LD + scol[j]) = ry[j];
            *reinterpret_cast<i32x4*>(base + TILE + srow[j] * F8T_LD + scol[j]) = rx[j];
        }
    };

    const int fi = lane & 15, fg = lane >> 4;
    const int tr_off = (8 * fg + (fi >> 1)) * F8T_LD + (fi & 1) * 8;      // this lane's 8 bytes of the 8 x 16 matrix of its group
    const int y_off = tr_off + wn * 128;                                   // + fn * 16 bytes ; + 32-row blocks
    const int x_off = TILE + tr_off + wk * 64;                             // + fk * 16 bytes

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nsteps = (mend - mbeg + BKM - 1) / BKM;
    if (nsteps > 0) {
        gload(mbeg);
        lstore(0);
        __syncthreads();
        for (int t = 0; t < nsteps; ++t) {
            if (t + 1 < nsteps) gload(mbeg + (t + 1) * BKM);
            const char* base = lds + (t & 1) * STAGE;
            if constexpr (KB == 32) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {          // 32 token rows each
                    long xf[4];
#pragma unroll
                    for (int f = 0; f < 4; ++f) xf[f] = __builtin_bit_cast(long, tr8(base + x_off + ks * 32 * F8T_LD + f * 16));
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        long yf[4];
#pragma unroll
                        for (int f = 0; f < 4; ++f) yf[f] = __builtin_bit_cast(long, tr8(base + y_off + ks * 32 * F8T_LD + (h * 4 + f) * 16));
#pragma unroll
                        for (int f = 0; f < 4; ++f)
#pragma unroll
                            for (int fk = 0; fk < 4; ++fk)
                                acc[h * 4 + f][fk] = __builtin_amdgcn_mfma_f32_16x16x32_bf8_fp8(yf[f], xf[fk], acc[h * 4 + f][fk], 0, 0, 0);
                    }
                }
            } else {
                auto frag128 = [&](int off) -> i32x8 {     // token rows q * 32 + 8 g .. + 7, q = 0 .. 3
                    const i32x2 a = tr8(base + off), b = tr8(base + off + 32 * F8T_LD), c = tr8(base + off + 64 * F8T_LD),
                                d = tr8(base + off + 96 * F8T_LD);
                    return i32x8{a[0], a[1], b[0], b[1], c[0], c[1], d[0], d[1]};
                };
                i32x8 xf[4];
#pragma unroll
                for (int f = 0; f < 4; ++f) xf[f] = frag128(x_off + f * 16);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    i32x8 yf[4];
#pragma unroll
                    for (int f = 0; f < 4; ++f) yf[f] = frag128(y_off + (h * 4 + f) * 16);
#pragma unroll
                    for (int f = 0; f < 4; ++f)
#pragma unroll
                        for (int fk = 0; fk < 4; ++fk)      // cbsz = 1: A (dY) is e5m2; blgp = 0: B (X) is e4m3; zero scale operands = unscaled form
                            acc[h * 4 + f][fk] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(yf[f], xf[fk], acc[h * 4 + f][fk], 1, 0, 0, 0, 0, 0);
                }
            }
            if (t + 1 < nsteps) lstore((t + 1) & 1);
            __syncthreads();
        }
    }
    float* out = ws + (long long)split * N * K;
#pragma unroll
    for (int fn = 0; fn < 8; ++fn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + wn * 128 + fn * 16 + 4 * fg + r;
            if (n >= N) continue;
#pragma unroll
            for (int fk = 0; fk < 4; ++fk) {
                const int k = k0 + wk * 64 + fk * 16 + fi;
                if (k < K) out[(long long)n * K + k] = acc[fn][fk][r];
            }
        }
}

// dW = (sum of the split slabs) * alpha_y * alpha_x (+ dW): the fold of vitk_gemm_tn_bf16 with the two inverse scales
template <typename OT>
__global__ __launch_bounds__(256) void tn_reduce_scaled_kernel(const float* __restrict__ ws, int splits, long long NK, int K,
                                                                OT* __restrict__ out, long long ldo, int accumulate,
                                                                const float* __restrict__ alpha_y, const float* __restrict__ alpha_x) {
    const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= NK) return;
    f32x4 s = *reinterpret_cast<const f32x4*>(ws + i4);
    for (int p = 1; p < splits; ++p) s += *reinterpret_cast<const f32x4*>(ws + (long long)p * NK + i4);
    s *= (alpha_y ? *alpha_y : 1.0f) * (alpha_x ? *alpha_x : 1.0f);
    const long long n = i4 / K, k = i4 % K;             // K % 4 == 0 -> the 4 elements share a row
    OT* o = out + n * ldo + k;
    if (accumulate) s += load4<OT>(o);
    store4<OT>(o, s);
}

template <typename Kern>
int f8t_set_max_lds(Kern kernel, int bytes) {
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

bool f8t_ok(int64_t M, int64_t N, int64_t K) { return M >= 1024 && N >= 256 && K >= 256 && (N % 16) == 0 && (K % 16) == 0 && M <= (1 << 30); }

}  // namespace

extern "C" int vitk_get_cu_reserve(void);

int64_t tn_pick_splits(int64_t tiles, int64_t M, int cus, long long slab_elems, double us_per_row, int64_t min_rows);      // gemm_bf16.hip

extern "C" int64_t vitk_gemm_tn_fp8_splits(int64_t M, int64_t N, int64_t K, int flags) {
    if (!f8t_ok(M, N, K)) return 0;
    const int64_t bkm = (flags & 1) ? 128 : 64;
    const int64_t tiles = ((N + 255) / 256) * ((K + 255) / 256);
    // rounds of (256 - reserve) one-CU jobs, like the 16-bit kernel (gemm_bf16.hip: tn_pick_splits); an fp8 job runs ~0.6 x the time per row
    const int64_t s = tn_pick_splits(tiles, M, 256 - vitk_get_cu_reserve(), (long long)N * K, 0.015, 4 * bkm);
    return s;
}

extern "C" int vitk_gemm_tn_fp8(const void* dY8, int64_t ldy, const void* X8, int64_t ldx, void* dW, int odt, int64_t ldo, int accumulate,
                                int64_t M, int64_t N, int64_t K, float* ws, int64_t splits, const float* alpha_y, const float* alpha_x,
                                int flags, void* stream) {
    if (!dY8 || !X8 || !dW || !ws) VITK_FAIL(VITK_E_ARG, "gemm_tn_fp8: null pointer");
    if (flags & ~1) VITK_FAIL(VITK_E_ARG, "gemm_tn_fp8: unknown flags 0x%x", (unsigned)flags);
    if (!f8t_ok(M, N, K) || splits < 1 || splits > 65535)
        VITK_FAIL(VITK_E_SHAPE, "gemm_tn_fp8: need M >= 1024, N, K >= 256 and multiples of 16 (M=%lld N=%lld K=%lld)", (long long)M, (long long)N, (long long)K);
    if ((ldy & 15) || (ldx & 15) || (ldo & 3) || !aligned16(dY8) || !aligned16(X8) || !aligned16(dW) || !aligned16(ws))
        VITK_FAIL(VITK_E_ALIGN, "gemm_tn_fp8: ldy / ldx %% 16, ldo %% 4 and 16-byte aligned pointers required");
    hipStream_t st = (hipStream_t)stream;
    const int tiles_n = (int)((N + 255) / 256), tiles_k = (int)((K + 255) / 256);
    const int nwg = tiles_n * tiles_k;
    const int bkm = (flags & 1) ? 128 : 64;
    long long rps = (M + splits - 1) / splits;
    rps = (rps + bkm - 1) / bkm * bkm;
    const int lds_bytes = 4 * bkm * F8T_LD;             // 2 stages x 2 operands
    if (flags & 1) {
        static const int rc__ = f8t_set_max_lds(gemm_tn256_f8_kernel<128>, 4 * 128 * F8T_LD);
        if (rc__ != 0) VITK_FAIL(rc__, "gemm_tn_fp8: cannot enable %d B of LDS", lds_bytes);
        hipLaunchKernelGGL(gemm_tn256_f8_kernel<128>, dim3((unsigned)(nwg * splits)), dim3(512), lds_bytes, st, (const unsigned char*)dY8,
                           (long long)ldy, (const unsigned char*)X8, (long long)ldx, ws, (int)M, (int)N, (int)K, (int)rps, tiles_k, nwg);
    } else {
        static const int rc__ = f8t_set_max_lds(gemm_tn256_f8_kernel<32>, 4 * 64 * F8T_LD);
        if (rc__ != 0) VITK_FAIL(rc__, "gemm_tn_fp8: cannot enable %d B of LDS", lds_bytes);
        hipLaunchKernelGGL(gemm_tn256_f8_kernel<32>, dim3((unsigned)(nwg * splits)), dim3(512), lds_bytes, st, (const unsigned char*)dY8,
                           (long long)ldy, (const unsigned char*)X8, (long long)ldx, ws, (int)M, (int)N, (int)K, (int)rps, tiles_k, nwg);
    }
    VITK_CHECK_LAUNCH("gemm_tn_fp8");
    const long long NK = (long long)N * K;
    const unsigned blocks = (unsigned)((NK / 4 + 255) / 256);
    VITK_DISPATCH_DT(odt, OT, hipLaunchKernelGGL((tn_reduce_scaled_kernel<OT>), dim3(blocks), dim3(256), 0, st, ws, (int)splits, NK, (int)K,
                                                  (OT*)dW, (long long)ldo, accumulate, alpha_y, alpha_x));
    VITK_CHECK_LAUNCH("gemm_tn_fp8 (fold)");
    return 0;
}
