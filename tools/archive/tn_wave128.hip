// tools/tn_wave128.hip -- EXPERIMENT prepared for round 4.  NEVER RUN when written (round 3's GPU minutes were spent): it compiles
// (resource report below), nothing else is claimed.  Not part of libvitk.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/tn_wave128.hip -o tools/tn_wave128.bin -ldl
//   tools/tn_wave128.bin [M N K]          (default 50432 3072 768; run from the repo root: A/B against vit_pytorch_amd/libvitk.so)
//
// Why: the weight-gradient GEMM dW[N,K] = dY[M,N]^T X[M,K] (9.9 ms of the 35.5-ms ViT-B/16 step) is bound by the issue of its
// transposing LDS reads.  Round 3 measured that directly: the fp8 TN kernel in its K = 32 form has the SAME MFMA cycles per
// token, tile, staging and split structure as the bf16 kernel and half the ds_read_*_tr_* instructions per MFMA -- and runs
// x1.4-1.55 ("Round 3, config 5").  Fewer transposing reads per MFMA in bf16 means a larger wave tile:
// 128 x 128 per wave needs 16 fragments per 64 MFMAs (0.25) where 128 x 64 needs 12 per 32 (0.375).  Its 64 accumulator tiles are
// 256 registers, so: FOUR waves (one per SIMD, 512 registers each on the unified file: accumulators in AGPRs), no ping-pong
// partner -- the wave overlaps its own transposing reads and LDS-DMA issue with its MFMAs by program order (one small group of
// reads after every four MFMAs, pinned with sched_barrier), fragments double-buffered across K-steps:
//
//   step t:  P1  32 MFMAs (x[0..7] x y[0..3] of step t)  ||  read y[4..7] of step t, issue the DMA of step t + 3
//            P2  wait for this wave's DMA of step t + 1, ONE barrier
//            P3  32 MFMAs (x[0..7] x y[4..7] of step t)  ||  refill x[0..7], y[0..3] in place with step t + 1
//
// Compile-time state (hipcc, ROCm 7.2): 128 VGPRs + 256 AGPRs (all 64 accumulator tiles), no scratch; per step 64 MFMAs, 32
// transposing reads, 8 LDS-DMA instructions, one barrier.
//
// LDS image, DMA pieces and zero page are gemm_tn_dma.hip's (vit_pytorch_amd/csrc): 4 stages of 32 token rows, 3 steps in flight.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int T_PIECE = 1088;                       // 1 KiB of data (2 rows x 512 B) + 64 B
constexpr int T_OPER_BYTES = 16 * T_PIECE;          // 32 rows of one operand
constexpr int T_STAGE_BYTES = 2 * T_OPER_BYTES;     // 34,816
constexpr int T_LDS_BYTES = 4 * T_STAGE_BYTES;      // 139,264

__device__ __attribute__((aligned(16))) char w_zero_page[512];

__device__ __forceinline__ int w_xcd_swizzle(int b, int nwg) {
    const int q = nwg / 8, r = nwg % 8;
    const int xcd = b % 8, idx = b / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
__device__ __forceinline__ int w_img(int row, int byte_in_row) {
    return (row >> 1) * T_PIECE + (row & 1) * 512 + (byte_in_row ^ ((row & 1) << 5));
}
__device__ __forceinline__ s16x4 w_tr(unsigned lds_addr) {
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(lds_addr) : "memory");
    return v;
}
__device__ __forceinline__ bf16x8 w_frag(unsigned stage, int off_lo, int off_hi) {
    const s16x4 lo = w_tr(stage + off_lo), hi = w_tr(stage + off_hi);
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
#define W_PIN() __builtin_amdgcn_sched_barrier(0)

__global__ __launch_bounds__(256) void gemm_tn_wave128_kernel(
    const __bf16* __restrict__ dY, long long ldy, const __bf16* __restrict__ X, long long ldx,
    float* __restrict__ ws, int M, int N, int K, int rows_per_split, int tiles_k, int nwg) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wk = wave & 1;        // wave tile: 128 (n) x 128 (k)
    const int lin = w_xcd_swizzle(blockIdx.x, (int)gridDim.x);
    const int split = lin / nwg;
    const int wg = lin % nwg;
    const int tn = wg / tiles_k, tk = wg % tiles_k;
    const int n0 = tn * 256, k0 = tk * 256;
    const int mbeg = split * rows_per_split;
    int mend = mbeg + rows_per_split; mend = mend < M ? mend : M;
    const int nsteps = mend > mbeg ? (mend - mbeg + 31) / 32 : 0;

    // producer: wave w fills pieces 4w .. 4w + 3 of each operand (piece p = token rows 2p, 2p + 1 of the step): 8 DMA per step.
    // Per-lane source pointers of step 0 are formed once; a step adds a wave-uniform stride.  EVERY step issues its 8 instructions
    // (past the end the last step is loaded again into its own stage: the same bytes), so the loop body has no branch around them
    // and the counted wait is always vmcnt(16).
    const int prow = lane >> 5;
    const int pchunk = (lane & 31) ^ (prow << 1);
    const bool y_ok = n0 + pchunk * 8 < N, x_ok = k0 + pchunk * 8 < K;
    const char* zsrc = w_zero_page + (lane & 31) * 16;
    const char* ysrc[4]; const char* xsrc[4];
    int mrow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        mrow[j] = mbeg + (wave * 4 + j) * 2 + prow;
        ysrc[j] = reinterpret_cast<const char*>(dY + (long long)mrow[j] * ldy + n0 + pchunk * 8);
        xsrc[j] = reinterpret_cast<const char*>(X + (long long)mrow[j] * ldx + k0 + pchunk * 8);
    }
    const long long ystep = 64LL * ldy, xstep = 64LL * ldx;        // bytes per 32 token rows
    const int last = nsteps - 1;
    auto issue2 = [&](int step_raw, int j, bool second) {        // two DMA instructions: pieces 4w + 2j, 4w + 2j + 1 of one operand
        const int step = step_raw < last ? step_raw : last;       // scalar
        char* base = lds + (step & 3) * T_STAGE_BYTES + (second ? T_OPER_BYTES : 0);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int q = 2 * j + jj;
            const bool mv = mrow[q] + step * 32 < mend;
            const char* src = second ? ((mv && x_ok) ? xsrc[q] + step * xstep : zsrc) : ((mv && y_ok) ? ysrc[q] + step * ystep : zsrc);
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                             (void __attribute__((address_space(3)))*)(base + (wave * 4 + q) * T_PIECE), 16, 0, 0);
        }
    };
    auto issue_all = [&](int step) { issue2(step, 0, false); issue2(step, 1, false); issue2(step, 0, true); issue2(step, 1, true); };

    // consumer: lane (fi, fg) reads row 4 fg + (fi >> 2) (and + 16), bytes (fi & 3) * 8 of a 32-byte block
    const int fi = lane & 15, fg = lane >> 4;
    const int r_lo = 4 * fg + (fi >> 2), r_hi = r_lo + 16;
    const int cb = (fi & 3) * 8;
    int y_lo[8], y_hi[8], x_lo[8], x_hi[8];
#pragma unroll
    for (int f = 0; f < 8; ++f) {
        y_lo[f] = w_img(r_lo, wn * 256 + f * 32 + cb); y_hi[f] = w_img(r_hi, wn * 256 + f * 32 + cb);
        x_lo[f] = T_OPER_BYTES + w_img(r_lo, wk * 256 + f * 32 + cb); x_hi[f] = T_OPER_BYTES + w_img(r_hi, wk * 256 + f * 32 + cb);
    }

    f32x4 acc[8][8];            // acc[fk][fn][j]: k = wk * 128 + 16 fk + 4 fg + j, n = wn * 128 + 16 fn + fi
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)lds);

    // one K-step; (XC, YAC): x[0..7] and y[0..3] of step t, in registers on entry; on exit they hold the same of step t + 1: every
    // fragment is refilled IN PLACE right after the last MFMA that reads it (y[0..3] are dead after P1, x[fk] after its four MFMAs of
    // P3).  ONE instance of this body in the loop: with two (fragment sets swapping roles) or three (+ tail) instances hipcc
    // spilled 100-160 registers in spite of a 128-register demand (tried: a lambda over array references keeps them in scratch).
#define W_STEP(T_, XC, YAC) do { \
        const int t_ = (T_); \
        const unsigned base = lds_base + (t_ & 3) * T_STAGE_BYTES; \
        const unsigned nbase = lds_base + ((t_ + 1) & 3) * T_STAGE_BYTES; \
        bf16x8 yb[4]; \
        /* ---- P1 ---- */ \
        __builtin_amdgcn_s_setprio(1); \
        _Pragma("unroll") for (int fk = 0; fk < 8; ++fk) { \
            _Pragma("unroll") for (int f = 0; f < 4; ++f) acc[fk][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(XC[fk], YAC[f], acc[fk][f], 0, 0, 0); \
            W_PIN(); \
            if (fk < 4) yb[fk] = w_frag(base, y_lo[4 + fk], y_hi[4 + fk]); \
            else issue2(t_ + 3, fk & 1, fk >= 6);   /* fk = 4, 5: dY pieces; 6, 7: X pieces -- stage (t + 3) & 3 was last read in step t - 1 */ \
            W_PIN(); \
        } \
        __builtin_amdgcn_s_setprio(0); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        W_PIN(); \
        /* ---- P2: this wave's DMA of step t + 1 has landed (the 16 instructions of steps t + 2, t + 3 may still fly) ---- */ \
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); \
        W_PIN(); \
        __builtin_amdgcn_s_barrier(); \
        W_PIN(); \
        /* ---- P3 (past the end the "next" fragments are read from a stage nobody uses any more and dropped) ---- */ \
        __builtin_amdgcn_s_setprio(1); \
        _Pragma("unroll") for (int fk = 0; fk < 8; ++fk) { \
            _Pragma("unroll") for (int f = 0; f < 4; ++f) acc[fk][4 + f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(XC[fk], yb[f], acc[fk][4 + f], 0, 0, 0); \
            W_PIN(); \
            XC[fk] = w_frag(nbase, x_lo[fk], x_hi[fk]); \
            if (fk < 4) YAC[fk] = w_frag(nbase, y_lo[fk], y_hi[fk]); \
            W_PIN(); \
        } \
        __builtin_amdgcn_s_setprio(0); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        W_PIN(); \
    } while (0)

    if (nsteps > 0) {
        issue_all(0); issue_all(1); issue_all(2);           // (clamped to the last step where the split is shorter)
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        W_PIN();
        __builtin_amdgcn_s_barrier();
        W_PIN();
        bf16x8 xa[8], ya[4];
#pragma unroll
        for (int f = 0; f < 8; ++f) xa[f] = w_frag(lds_base, x_lo[f], x_hi[f]);
#pragma unroll
        for (int f = 0; f < 4; ++f) ya[f] = w_frag(lds_base, y_lo[f], y_hi[f]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        W_PIN();
        for (int t = 0; t < nsteps; ++t) W_STEP(t, xa, ya);
#undef W_STEP
    }

    // partial tile -> ws[split][n][k]: 16 bytes per lane (4 consecutive k)
    float* out = ws + (long long)split * N * K;
#pragma unroll
    for (int fn = 0; fn < 8; ++fn) {
        const int n = n0 + wn * 128 + fn * 16 + fi;
        if (n >= N) continue;
#pragma unroll
        for (int fk = 0; fk < 8; ++fk) {
            const int k = k0 + wk * 128 + fk * 16 + 4 * fg;
            if (k < K) *reinterpret_cast<f32x4*>(out + (long long)n * K + k) = acc[fk][fn];
        }
    }
}

__global__ void w_reduce_kernel(const float* ws, int splits, long long NK, float* out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= NK) return;
    float s = 0.f;
    for (int p = 0; p < splits; ++p) s += ws[(long long)p * NK + i];
    out[i] = s;
}
__global__ void w_fill_kernel(__bf16* p, long long n, unsigned seed, float scale) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned)i * 2654435761u ^ seed; x ^= x >> 16; x *= 0x21f0aaadu; x ^= x >> 15; x *= 0x735a2d97u; x ^= x >> 15;
    p[i] = (__bf16)(((float)(x & 0xffff) / 32768.0f - 1.0f) * scale * (1.0f + (float)(i % 7) * 0.25f));     // asymmetric in both indices
}

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e__), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const long long M = argc > 3 ? atoll(argv[1]) : 50432, N = argc > 3 ? atoll(argv[2]) : 3072, K = argc > 3 ? atoll(argv[3]) : 768;
    if ((N & 7) || (K & 7)) { printf("N, K must be multiples of 8\n"); return 1; }
    void* lib = dlopen("vit_pytorch_amd/libvitk.so", RTLD_NOW);
    typedef int64_t (*splits_fn)(int64_t, int64_t, int64_t);
    typedef int (*tn_fn)(const void*, int64_t, const void*, int64_t, void*, int, int64_t, int, int64_t, int64_t, int64_t, float*, int64_t, void*);
    splits_fn vitk_splits = lib ? (splits_fn)dlsym(lib, "vitk_gemm_tn_splits") : nullptr;
    tn_fn vitk_tn = lib ? (tn_fn)dlsym(lib, "vitk_gemm_tn_bf16") : nullptr;
    if (!vitk_splits || !vitk_tn) { printf("libvitk.so not found (run from the repo root after building)\n"); return 1; }
    const long long splits = vitk_splits(M, N, K);
    const int tiles_n = (int)((N + 255) / 256), tiles_k = (int)((K + 255) / 256), nwg = tiles_n * tiles_k;
    long long rps = (M + splits - 1) / splits; rps = (rps + 31) / 32 * 32;
    __bf16 *dY, *X; float *ws, *out_new, *out_ref;
    CK(hipMalloc(&dY, M * N * 2)); CK(hipMalloc(&X, M * K * 2)); CK(hipMalloc(&ws, splits * N * K * 4));
    CK(hipMalloc(&out_new, N * K * 4)); CK(hipMalloc(&out_ref, N * K * 4));
    w_fill_kernel<<<(unsigned)((M * N + 255) / 256), 256>>>(dY, M * N, 1u, 0.02f);
    w_fill_kernel<<<(unsigned)((M * K + 255) / 256), 256>>>(X, M * K, 2u, 1.0f);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_wave128_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS_BYTES));
    auto run_new = [&]() {
        hipLaunchKernelGGL(gemm_tn_wave128_kernel, dim3((unsigned)(nwg * splits)), dim3(256), T_LDS_BYTES, 0, dY, N, X, K, ws, (int)M, (int)N, (int)K,
                           (int)rps, tiles_k, nwg);
        w_reduce_kernel<<<(unsigned)((N * K + 255) / 256), 256>>>(ws, (int)splits, N * K, out_new);
    };
    auto run_ref = [&]() { return vitk_tn(dY, N, X, K, out_ref, /*f32*/ 0, K, 0, M, N, K, ws, splits, nullptr); };
    if (run_ref() != 0) { printf("vitk_gemm_tn_bf16 failed\n"); return 1; }
    run_new();
    CK(hipDeviceSynchronize());
    std::vector<float> a((size_t)(N * K)), b((size_t)(N * K));
    CK(hipMemcpy(a.data(), out_new, N * K * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), out_ref, N * K * 4, hipMemcpyDeviceToHost));
    double num = 0, den = 0;
    for (size_t i = 0; i < a.size(); ++i) { num += ((double)a[i] - b[i]) * ((double)a[i] - b[i]); den += (double)b[i] * b[i]; }
    const double rel = den > 0 ? sqrt(num / den) : -1;
    printf("M=%lld N=%lld K=%lld splits=%lld: rel. error vs the production kernel %.3e  %s\n", M, N, K, splits, rel, rel < 1e-5 ? "NUMERICS_OK" : "NUMERICS_DIFFERENT");
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms_new = 0, ms_ref = 0;
    for (int rep = 0; rep < 2; ++rep) {        // interleaved A/B, second round reported
        CK(hipEventRecord(e0)); for (int i = 0; i < 20; ++i) run_new(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_new, e0, e1));
        CK(hipEventRecord(e0)); for (int i = 0; i < 20; ++i) run_ref(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_ref, e0, e1));
    }
    const double fl = 2.0 * M * N * K;
    printf("wave128 kernel (+ its fold): %.3f ms = %.0f TF/s;  production vitk_gemm_tn_bf16: %.3f ms = %.0f TF/s\n", ms_new / 20, fl / (ms_new / 20) / 1e9,
           ms_ref / 20, fl / (ms_ref / 20) / 1e9);
    return 0;
}
