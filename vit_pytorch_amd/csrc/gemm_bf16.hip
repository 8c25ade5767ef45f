// gemm_bf16.hip -- bf16 MFMA GEMMs for gfx950 (MI355X).
//
// Replace nn.Linear (vit.py:20,23,44,47,102; simple_vit.py:30,32,47,48,93) forward and autograd.
//
//   NT  C[M,N]  = A[M,K] . W[N,K]^T   (+ fused epilogue)      forward Linear, and dX with W^T
//   TN  dW[N,K] = dY[M,N]^T . X[M,K]  (split over M)          weight gradients
//
// Kernels (all f32 accumulation on v_mfma_f32_16x16x32_bf16; the f16 library flavour uses ..._f16, see common.h):
//  * gemm_nt256pp_kernel<EPI, FM>: the production NT kernel -- 256(224) x 256 tile, 8 waves, four LDS-DMA stages,
//    ping-pong wave groups (described at the kernel).  Used when M >= 1024, N >= 256, K % 64 == 0.
//  * gemm_nt_kernel<EPI>: 128 x 128 tile, 4 waves (2x2, 64x64 per wave), K-step 32, two LDS-DMA stages, one barrier per
//    K-step: small and odd shapes (the classifier head, tiny models).
//  * gemm_tn256_kernel / gemm_tn_kernel + tn_reduce_kernel: weight gradients.  The reduction index (token row m) is the
//    strided one in both operands; tiles are staged row-major through registers into padded LDS rows and read with
//    ds_read_b64_tr_b16 (hardware transpose read).  Split-M slabs of f32 partials + a deterministic reduce (no atomics),
//    which also converts to the gradient dtype / accumulates.
// Common to the NT kernels:
//  * operands go HBM -> LDS with global_load_lds dwordx4 (no VGPR round trip).  The LDS image is lane-linear (DMA
//    constraint), so the bank swizzle is applied to the per-lane SOURCE address and to the ds_read_b128 address (the same
//    involution on both sides): SQ_LDS_BANK_CONFLICT = 0.
//  * the MFMA "A" operand is the W fragment and "B" the activation fragment, so a lane ends up holding 4 CONSECUTIVE output
//    columns of one output row; bias / GELU / residual / GELU' are applied in the epilogue and the activation tensor is
//    written exactly once.
//  * blockIdx -> tile mapping is XCD-aware: each of the 8 XCDs walks a contiguous run of tiles, so the activation panel
//    a tile row shares is fetched into one L2, not eight.
#include "common.h"
#include <atomic>
#include "gemm_nt_plan.h"
#include <stdlib.h>

namespace {

constexpr int BM = 128, BN = 128;
constexpr int NT_BK = 32;
constexpr int NT_TILE_BYTES = 128 * NT_BK * 2;           // 8 KiB per operand tile
constexpr int NT_STAGE_BYTES = 2 * NT_TILE_BYTES;        // A + W
constexpr int NXCD = 8;

__device__ __forceinline__ int xcd_swizzle(int b, int nwg) {
    const int q = nwg / NXCD, r = nwg % NXCD;
    const int xcd = b % NXCD, idx = b / NXCD;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Grouped tile order: walk the n-tiles in groups of `gn` (m fastest across the group, n fastest inside it), so
// the ~32 tiles an XCD runs at once cover (32/gn) activation panels x gn weight panels instead of ~3 x all.
// Fabric bytes ~ A_total * tiles_n / gn + W_panel * gn * tiles / 32: for N = 3072, K = 768 (W = 4.7 MB > L2)
// gn = 6 reads 328 MB instead of 425 MB (measured 485 MB with the plain n-fastest order).
__device__ __forceinline__ void grouped_tile(int wg, int tiles_m, int tiles_n, int gn, int& tm, int& tn) {
    const int per_group = gn * tiles_m;
    const int g = wg / per_group;
    const int r = wg - g * per_group;
    const int rem = tiles_n - g * gn;
    const int w = rem < gn ? rem : gn;
    tm = r / w;
    tn = g * gn + (r - tm * w);
}

// permutation f = [0,2,3,1]: logical 16-byte chunk g of row i sits at position g ^ f[(i>>2)&3].
// Makes every ds_read_b128 lane group of the 64-byte-row image hit 16 distinct bank slots.
__device__ __forceinline__ int swz_f(int x) { return (0x1320 >> (4 * (x & 3))) & 3; }

template <int EPI>
__global__ __launch_bounds__(256) void gemm_nt_kernel(
    const __bf16* __restrict__ A, long long lda, const __bf16* __restrict__ W, long long ldw,
    void* __restrict__ Cv, long long ldc, int M, int N, int K,
    const __bf16* __restrict__ bias, const float* __restrict__ resid, __bf16* __restrict__ aux,
    int tiles_n, int nwg) {
    __shared__ __attribute__((aligned(16))) char lds[2 * NT_STAGE_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int wg = xcd_swizzle(blockIdx.x, nwg);
    const int tm = wg / tiles_n, tn = wg % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- staging addresses (constant over the K loop except the k offset) ----
    const int srow = lane >> 2;                 // row inside a 16-row group
    const int spos = lane & 3;                  // 16-byte slot inside the 64-byte row
    const int schunk = spos ^ swz_f(lane >> 4); // logical k-chunk this lane fetches
    const __bf16* a_src[2];
    const __bf16* w_src[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rg = wave * 2 + j;
        int ar = m0 + rg * 16 + srow; ar = ar < M ? ar : M - 1;
        int wr = n0 + rg * 16 + srow; wr = wr < N ? wr : N - 1;
        a_src[j] = A + (long long)ar * lda + schunk * 8;
        w_src[j] = W + (long long)wr * ldw + schunk * 8;
    }
    auto stage = [&](int buf, int kt) {
        char* base = lds + buf * NT_STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rg = wave * 2 + j;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(a_src[j] + kt * NT_BK),
                                             (void __attribute__((address_space(3)))*)(base + rg * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(w_src[j] + kt * NT_BK),
                                             (void __attribute__((address_space(3)))*)(base + NT_TILE_BYTES + rg * 1024), 16, 0, 0);
        }
    };

    // ---- fragment read addresses ----
    const int fi = lane & 15, fg = lane >> 4;
    const int fpos = fg ^ swz_f(fi >> 2);
    const int a_off = (wm * 64 + fi) * 64 + fpos * 16;                   // + fm*16*64
    const int w_off = NT_TILE_BYTES + (wn * 64 + fi) * 64 + fpos * 16;   // + fn*16*64

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nt = K / NT_BK;
    stage(0, 0);
    for (int t = 0; t < nt; ++t) {
        __syncthreads();  // own DMA landed (vmcnt(0)) + everyone's; also: all reads of the other stage are done
        if (t + 1 < nt) stage((t + 1) & 1, t + 1);
        const char* base = lds + (t & 1) * NT_STAGE_BYTES;
        bf16x8 xf[4], wf[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            xf[f] = *reinterpret_cast<const bf16x8*>(base + a_off + f * 1024);
            wf[f] = *reinterpret_cast<const bf16x8*>(base + w_off + f * 1024);
        }
#pragma unroll
        for (int fn = 0; fn < 4; ++fn)
#pragma unroll
            for (int fm = 0; fm < 4; ++fm)
                acc[fn][fm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[fn], xf[fm], acc[fn][fm], 0, 0, 0);
    }

    // ---- epilogue: lane holds C[m][n..n+3], m = ..+(lane&15), n = ..+4*(lane>>4) ----
#pragma unroll
    for (int fm = 0; fm < 4; ++fm) {
        const int m = m0 + wm * 64 + fm * 16 + fi;
        if (m >= M) continue;
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) {
            const int n = n0 + wn * 64 + fn * 16 + 4 * fg;
            if (n >= N) continue;
            f32x4 v = acc[fn][fm];
            const long long o = (long long)m * ldc + n;
            if constexpr (EPI == VITK_EPI_BIAS || EPI == VITK_EPI_BIAS_GELU) {
                v += load4<__bf16>(bias + n);
            }
            if constexpr (EPI == VITK_EPI_NONE || EPI == VITK_EPI_BIAS) {
                store4<__bf16>(reinterpret_cast<__bf16*>(Cv) + o, v);
            } else if constexpr (EPI == VITK_EPI_BIAS_GELU) {
                store4<__bf16>(aux + o, v);
                f32x4 gq;
#pragma unroll
                for (int e = 0; e < 4; ++e) gq[e] = gelu_erf(v[e]);
                store4<__bf16>(reinterpret_cast<__bf16*>(Cv) + o, gq);
            } else if constexpr (EPI == VITK_EPI_RESID) {
                if (bias) v += load4<__bf16>(bias + n);
                v += *reinterpret_cast<const f32x4*>(resid + o);
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(Cv) + o) = v;
            } else if constexpr (EPI == VITK_EPI_GELU_BWD) {
                const f32x4 h = load4<__bf16>(aux + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= gelu_erf_grad(h[e]);
                store4<__bf16>(reinterpret_cast<__bf16*>(Cv) + o, v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// TN: dW[n][k] = sum_m dY[m][n] * X[m][k]
// ------------------------------------------------------------------------------------------
constexpr int TN_BKM = 32;                 // token rows per step
constexpr int TN_LD = 288;                 // bytes per LDS row: 256 data + 32 pad (8 consecutive rows -> 64 distinct banks)
constexpr int TN_TILE_BYTES = TN_BKM * TN_LD;
constexpr int TN_STAGE_BYTES = 2 * TN_TILE_BYTES;

__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int off) {
    // two transpose reads: token rows {4g..4g+3} and {16+4g..16+4g+3} of one 16-column block
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(tile + off));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(tile + off + 16 * TN_LD));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(256) void gemm_tn_kernel(
    const __bf16* __restrict__ dY, long long ldy, const __bf16* __restrict__ X, long long ldx,
    float* __restrict__ ws, int M, int N, int K, int rows_per_split, int tiles_k, int nwg) {
    __shared__ __attribute__((aligned(16))) char lds[2 * TN_STAGE_BYTES];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wn = wave >> 1, wk = wave & 1;
    const int lin = xcd_swizzle(blockIdx.x, (int)gridDim.x);   // (split, tile) jointly, split-major (see gemm_tn256_kernel)
    const int split = lin / nwg;
    const int wg = lin % nwg;
    const int tn = wg / tiles_k, tk = wg % tiles_k;
    const int n0 = tn * BN, k0 = tk * BM;
    const int mbeg = split * rows_per_split;
    int mend = mbeg + rows_per_split; mend = mend < M ? mend : M;

    // staging: 512 16-byte chunks per operand tile, 2 per thread
    int srow[2], scol[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { const int c = tid + 256 * j; srow[j] = c >> 4; scol[j] = (c & 15) * 8; }
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    bf16x8 ry[2], rx[2];
    auto gload = [&](int mb) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = mb + srow[j];
            const bool mv = m < mend;
            ry[j] = (mv && n0 + scol[j] < N) ? *reinterpret_cast<const bf16x8*>(dY + (long long)m * ldy + n0 + scol[j]) : zero8;
            rx[j] = (mv && k0 + scol[j] < K) ? *reinterpret_cast<const bf16x8*>(X + (long long)m * ldx + k0 + scol[j]) : zero8;
        }
    };
    auto lstore = [&](int buf) {
        char* base = lds + buf * TN_STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            *reinterpret_cast<bf16x8*>(base + srow[j] * TN_LD + scol[j] * 2) = ry[j];
            *reinterpret_cast<bf16x8*>(base + TN_TILE_BYTES + srow[j] * TN_LD + scol[j] * 2) = rx[j];
        }
    };

    const int fi = lane & 15, fg = lane >> 4;
    // transpose-read address of this lane inside a 16-column block: row 4g + (i>>2), 4 columns at (i&3)*4
    const int tr_off = (4 * fg + (fi >> 2)) * TN_LD + (fi & 3) * 8;
    const int y_off = tr_off + wn * 128;                    // + fn*32 bytes
    const int x_off = TN_TILE_BYTES + tr_off + wk * 128;    // + fk*32 bytes

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nsteps = (mend - mbeg + TN_BKM - 1) / TN_BKM;
    if (nsteps > 0) {
        gload(mbeg);
        lstore(0);
        __syncthreads();
        for (int t = 0; t < nsteps; ++t) {
            if (t + 1 < nsteps) gload(mbeg + (t + 1) * TN_BKM);
            const char* base = lds + (t & 1) * TN_STAGE_BYTES;
            bf16x8 yf[4], xf[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                yf[f] = tr_frag(base, y_off + f * 32);
                xf[f] = tr_frag(base, x_off + f * 32);
            }
#pragma unroll
            for (int fn = 0; fn < 4; ++fn)
#pragma unroll
                for (int fk = 0; fk < 4; ++fk)
                    acc[fn][fk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[fn], xf[fk], acc[fn][fk], 0, 0, 0);
            if (t + 1 < nsteps) lstore((t + 1) & 1);
            __syncthreads();
        }
    }
    // partial tile -> ws[split][n][k]; D row = n (4g + r), D col = k (lane & 15)
    float* out = ws + (long long)split * N * K;
#pragma unroll
    for (int fn = 0; fn < 4; ++fn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + wn * 64 + fn * 16 + 4 * fg + r;
            if (n >= N) continue;
#pragma unroll
            for (int fk = 0; fk < 4; ++fk) {
                const int k = k0 + wk * 64 + fk * 16 + fi;
                if (k < K) out[(long long)n * K + k] = acc[fn][fk][r];
            }
        }
}

template <typename OT>
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ ws, int splits, long long NK, int K,
                                                         OT* __restrict__ out, long long ldo, int accumulate, long long slab_stride) {
    const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= NK) return;
    f32x4 s = *reinterpret_cast<const f32x4*>(ws + i4);
    for (int p = 1; p < splits; ++p) s += *reinterpret_cast<const f32x4*>(ws + (long long)p * slab_stride + i4);
    const long long n = i4 / K, k = i4 % K;  // K % 4 == 0 -> the 4 elements share a row
    OT* o = out + n * ldo + k;
    if (accumulate) s += load4<OT>(o);
    store4<OT>(o, s);
}


// ==========================================================================================
// 256 x 256 tiles, 8 waves (2 x 4), one workgroup per CU (128 KiB LDS).
// Per MFMA the 128^2 tile moves ~2x the LDS bytes (both DMA writes and fragment reads) and is
// LDS-bound on gfx950; a 128x64 wave tile needs 12 ds_read_b128 per 32 MFMAs instead of 8 per 16.
// ==========================================================================================
constexpr int L_BN = 256, L_BK = 64;

// ==========================================================================================
// Ping-pong variant of the 256 x 256 NT kernel.
// The two waves that share a SIMD (wave w and w+4) are put in different GROUPS that run the same
// instruction stream one barrier-slot apart: while group A issues its 16 MFMAs of a sub-step, group B
// reads its fragments from LDS and issues LDS-DMA for a future K-step, and vice versa.  The MFMA pipe of
// every SIMD then always has exactly one wave feeding it instead of two waves reading LDS together and
// then queueing on the pipe together.  K-step 32, FOUR 32 KiB stages: the DMA of K-step j+3 is issued
// during K-step j (counted vmcnt: never drained inside the loop).
//   slot (barrier interval) c:   4j     4j+1    4j+2    4j+3    4j+4
//   group A:                    R0(j)   M0(j)   R1(j)   M1(j)   R0(j+1)
//   group B:                    M1(j-1) R0(j)   M0(j)   R1(j)   M1(j)
// R0 = read W(4) + X[0..3](4) fragments, issue 2 DMA; R1 = read X[4..7], issue 2 DMA, counted wait for
// K-step j+1; M0/M1 = 16 MFMAs each.  LDS hazards: a stage is re-filled (K-step j+3 -> stage (j-1)&3) only
// after the barrier that follows the last read of K-step j-1 (reads are retired with lgkmcnt(0) BEFORE
// the barrier that ends an R slot); a stage is read only after every wave's counted vmcnt + a barrier.
// (A coarser split -- one R and one M slot per K-step, 12 reads / 32 MFMAs -- measured 3-5% slower.)
// ==========================================================================================
constexpr int P_BK = 32, P_STAGES = 4;
constexpr int P_TILE_BYTES = 256 * P_BK * 2;          // 16 KiB per operand per stage (64-byte rows)
constexpr int P_STAGE_BYTES = 2 * P_TILE_BYTES;       // 32 KiB
constexpr int P_LDS_BYTES = P_STAGES * P_STAGE_BYTES; // 128 KiB

#define PP_BARRIER() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

// EB = bytes per operand element: 2 (bf16 / f16) or 1 (fp8 e4m3, OCP): the 64-byte LDS rows then hold 64 k-values, every
// 16-byte fragment read feeds TWO v_mfma_f32_16x16x32_fp8_fp8 (its low and high 8 bytes: any split of the k index is legal
// as long as both operands use the same one), so a K-step covers 64 k with the same DMA / LDS traffic and barrier count as
// 32 k of bf16 -- the main loop is paced by exactly those.  `alpha` (fp8: 1 / (scale_a * scale_w)) multiplies the sums.
// F8 (EB == 1 only) -- bit 0: the A operand (activations / gradients) is OCP e5m2 ("bf8"), W stays e4m3: the backward's
// dX = dY . W with e5m2 gradients runs on v_mfma_f32_16x16x32_fp8_bf8.  Bit 1: the K = 128 instruction
// (v_mfma_f32_16x16x128_f8f6f4, unit block scales: twice the matrix rate of the K = 32 forms on gfx950): the fragments of
// an EVEN K-step are held in registers and multiplied together with those of the following odd K-step -- a lane then
// feeds 32 bytes per operand, and since both operands pair the same two K-steps in the same order the k permutation
// is a legal one.  DMA ring, LDS layout, barrier slots and epilogues are those of the K = 32 loop; needs K %% 128 == 0.
template <int EPI, int FM, int EB = 2, int F8 = 0>
__global__ __launch_bounds__(512) void gemm_nt256pp_kernel(
    const void* __restrict__ Av, long long lda, const void* __restrict__ Wv, long long ldw,
    void* __restrict__ Cv, long long ldc, int M, int N, int K,
    const __bf16* __restrict__ bias, const float* __restrict__ resid, __bf16* __restrict__ aux,
    int tiles_n, int nwg, int group_n, float* __restrict__ csum, float alpha, unsigned drop_t, unsigned drop_seed, float inv_keep,
    const float* __restrict__ alpha_a, const float* __restrict__ alpha_w, F8Out f8) {
    const char* A = reinterpret_cast<const char*>(Av);
    const char* W = reinterpret_cast<const char*>(Wv);
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const bool grp_b = wave >= 4;
    const int wg = xcd_swizzle(blockIdx.x, nwg);
    int tm, tn;
    grouped_tile(wg, nwg / tiles_n, tiles_n, group_n, tm, tn);
    constexpr int WROWS = 16 * FM;            // rows per wave (FM = 8: 256-row tile, FM = 7: 224-row tile)
    const int m0 = tm * (2 * WROWS), n0 = tn * L_BN;

    // staging: 64-byte rows, a wave-instruction fills 16 rows; wave w owns row groups 2w, 2w+1 of each operand
    const int srow = lane >> 2, spos = lane & 3;
    const int schunk = spos ^ swz_f(lane >> 4);
    const char* a_src[2];     // byte addresses: a K-step is 64 bytes of every row whatever the element size
    const char* w_src[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int ar = m0 + (wave * 2 + j) * 16 + srow; ar = ar < M ? ar : M - 1;
        int wr = n0 + (wave * 2 + j) * 16 + srow; wr = wr < N ? wr : N - 1;
        a_src[j] = A + (long long)ar * lda * EB + schunk * 16;
        w_src[j] = W + (long long)wr * ldw * EB + schunk * 16;
    }
    auto stage_a = [&](int kt) {
        char* base = lds + (kt & 3) * P_STAGE_BYTES + wave * 2048;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(a_src[j] + kt * 64),
                                             (void __attribute__((address_space(3)))*)(base + j * 1024), 16, 0, 0);
    };
    auto stage_w = [&](int kt) {
        char* base = lds + (kt & 3) * P_STAGE_BYTES + P_TILE_BYTES + wave * 2048;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(w_src[j] + kt * 64),
                                             (void __attribute__((address_space(3)))*)(base + j * 1024), 16, 0, 0);
    };
    auto mma = [&](const bf16x8& wfr, const bf16x8& xfr, f32x4 c) -> f32x4 {
        if constexpr (EB == 2) {
            return __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr, xfr, c, 0, 0, 0);
        } else {
            typedef long l2 __attribute__((ext_vector_type(2)));
            const l2 w2 = __builtin_bit_cast(l2, wfr), x2 = __builtin_bit_cast(l2, xfr);
            if constexpr (F8 & 1) {     // W (MFMA operand A) e4m3, activations / gradients (operand B) e5m2
                c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(w2[0], x2[0], c, 0, 0, 0);
                return __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(w2[1], x2[1], c, 0, 0, 0);
            } else {
                c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(w2[0], x2[0], c, 0, 0, 0);
                return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(w2[1], x2[1], c, 0, 0, 0);
            }
        }
    };
    // K = 128: (w0 | w1) . (x0 | x1) over two K-steps; zero scale operands select the unscaled instruction (scale 1.0)
    auto mma128 = [&](const bf16x8& w0, const bf16x8& w1, const bf16x8& x0, const bf16x8& x1, f32x4 c) -> f32x4 {
        typedef int i4 __attribute__((ext_vector_type(4)));
        typedef int i8 __attribute__((ext_vector_type(8)));
        const i8 a = __builtin_shufflevector(__builtin_bit_cast(i4, w0), __builtin_bit_cast(i4, w1), 0, 1, 2, 3, 4, 5, 6, 7);
        const i8 b = __builtin_shufflevector(__builtin_bit_cast(i4, x0), __builtin_bit_cast(i4, x1), 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0 /* A: e4m3 */, (F8 & 1) ? 1 : 0 /* B: e5m2 / e4m3 */, 0, 0, 0, 0);
    };
    static_assert(F8 == 0 || EB == 1, "the e5m2 / K = 128 flavours are fp8 flavours");

    const int fi = lane & 15, fg = lane >> 4;
    const int fpos = fg ^ swz_f(fi >> 2);
    const int a_off = (wm * WROWS + fi) * 64 + fpos * 16;                  // + fm * 1024
    const int w_off = P_TILE_BYTES + (wn * 64 + fi) * 64 + fpos * 16;      // + fn * 1024

    f32x4 acc[4][FM];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nt = K * EB / 64;
    // prologue: K-steps 0..2 in flight, K-step 0 landed
    stage_a(0); stage_w(0);
    if (nt > 1) { stage_a(1); stage_w(1); }
    if (nt > 2) { stage_a(2); stage_w(2); }
    if (nt > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (nt > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PP_BARRIER();
    if (grp_b) PP_BARRIER();   // group B runs one slot behind group A

    if constexpr ((F8 & 2) != 0) {
        // K = 128 flavour: the slots of the loop below, K-steps taken in pairs -- even: fragments into hw / hx, no MFMA;
        // odd: 16 MFMAs of K = 128 per M slot (the cycles of the 32 K = 32 ones they replace, covering two K-steps)
        bf16x8 hw[4], hx[FM];
        for (int j = 0; j < nt; ++j) {
            const char* base = lds + (j & 3) * P_STAGE_BYTES;
            const bool more = j + 3 < nt;
            const bool odd = (j & 1) != 0;          // wave-uniform
            bf16x8 wf[4], xf[4];
            // ---- R0 ----
#pragma unroll
            for (int f = 0; f < 4; ++f) wf[f] = *reinterpret_cast<const bf16x8*>(base + w_off + f * 1024);
#pragma unroll
            for (int f = 0; f < 4; ++f) xf[f] = *reinterpret_cast<const bf16x8*>(base + a_off + f * 1024);
            if (more) stage_a(j + 3);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PP_BARRIER();
            // ---- M0 ----
            if (odd) {
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int fn = 0; fn < 4; ++fn)
#pragma unroll
                    for (int f = 0; f < 4; ++f)
                        acc[fn][f] = mma128(hw[fn], wf[fn], hx[f], xf[f], acc[fn][f]);
                __builtin_amdgcn_s_setprio(0);
            } else {
#pragma unroll
                for (int f = 0; f < 4; ++f) { hw[f] = wf[f]; hx[f] = xf[f]; }
            }
            PP_BARRIER();
            // ---- R1 ----
#pragma unroll
            for (int f = 0; f < FM - 4; ++f) xf[f] = *reinterpret_cast<const bf16x8*>(base + a_off + (4 + f) * 1024);
            if (more) stage_w(j + 3);
            if (more) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (j + 2 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PP_BARRIER();
            // ---- M1 ----
            if (odd) {
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int fn = 0; fn < 4; ++fn)
#pragma unroll
                    for (int f = 0; f < FM - 4; ++f)
                        acc[fn][4 + f] = mma128(hw[fn], wf[fn], hx[4 + f], xf[f], acc[fn][4 + f]);
                __builtin_amdgcn_s_setprio(0);
            } else {
#pragma unroll
                for (int f = 0; f < FM - 4; ++f) hx[4 + f] = xf[f];
            }
            PP_BARRIER();
        }
    } else
    for (int j = 0; j < nt; ++j) {
        const char* base = lds + (j & 3) * P_STAGE_BYTES;
        const bool more = j + 3 < nt;
        bf16x8 wf[4], xf[4];
        // ---- R0 ----
#pragma unroll
        for (int f = 0; f < 4; ++f) wf[f] = *reinterpret_cast<const bf16x8*>(base + w_off + f * 1024);
#pragma unroll
        for (int f = 0; f < 4; ++f) xf[f] = *reinterpret_cast<const bf16x8*>(base + a_off + f * 1024);
        if (more) stage_a(j + 3);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PP_BARRIER();
        // ---- M0 ----
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int fn = 0; fn < 4; ++fn)
#pragma unroll
            for (int f = 0; f < 4; ++f)
                acc[fn][f] = mma(wf[fn], xf[f], acc[fn][f]);
        __builtin_amdgcn_s_setprio(0);
        PP_BARRIER();
        // ---- R1 ----
#pragma unroll
        for (int f = 0; f < FM - 4; ++f) xf[f] = *reinterpret_cast<const bf16x8*>(base + a_off + (4 + f) * 1024);
        if (more) stage_w(j + 3);
        // this wave's DMA of K-step j+1 must have landed before the barrier that precedes its first read
        if (more) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (j + 2 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PP_BARRIER();
        // ---- M1 ----
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int fn = 0; fn < 4; ++fn)
#pragma unroll
            for (int f = 0; f < FM - 4; ++f)
                acc[fn][4 + f] = mma(wf[fn], xf[f], acc[fn][4 + f]);
        __builtin_amdgcn_s_setprio(0);
        PP_BARRIER();
    }
    if (!grp_b) PP_BARRIER();  // pairs with group B's extra barrier
    if constexpr (EB == 1) {
        // per-tensor scales are undone here: alpha (host) x the two device-resident inverse scales (delayed scaling)
        const float al = alpha * (alpha_a ? *alpha_a : 1.0f) * (alpha_w ? *alpha_w : 1.0f);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < FM; ++jj) acc[i][jj] *= al;
    }

    // ---- epilogue: accumulators -> (wave-private 16 KiB of the now idle LDS) -> full-line global I/O ----
    // A lane holds 4 consecutive columns of 4*FM scattered (row, 16-col block) pairs; stored directly that is
    // 32-byte pieces of 16 rows per instruction (measured: ~30% of the kernel).  Re-staged through LDS every
    // global store (and every residual / pre-activation load) is 16 B per lane and 128 B contiguous per row.
    __syncthreads();  // every wave is done reading the operand stages
    char* ep = lds + wave * 16384;
    const int mrow0 = m0 + wm * WROWS, ncol0 = n0 + wn * 64;
    if constexpr (EPI == VITK_EPI_RESID) {
        const int rr = lane >> 4, rc = lane & 15;      // read-back: 4 rows x 16 chunks of 4 floats
        const int ncol = ncol0 + rc * 4;
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (bias && ncol < N) b4 = load4<__bf16>(bias + ncol);
        // the residual rows are fetched up front (16 x 16 B per lane and half) so their latency hides behind the
        // LDS staging instead of being paid once per row group
        constexpr int NF0 = 4, NF1 = FM - 4;
        f32x4 rpre[2][16];
        auto fetch_resid = [&](int hh) {
            const int nf = hh ? NF1 : NF0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int m = mrow0 + hh * 64 + j * 4 + rr;
                rpre[hh][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (j < nf * 4 && m < M && ncol < N) rpre[hh][j] = *reinterpret_cast<const f32x4*>(resid + (long long)m * ldc + ncol);
            }
        };
        fetch_resid(0);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int nf = hh ? NF1 : NF0;
#pragma unroll
            for (int f4 = 0; f4 < 4; ++f4) {
                if (f4 < nf) {
                    const int row = f4 * 16 + fi;
#pragma unroll
                    for (int fn = 0; fn < 4; ++fn) {
                        const int c16 = fn * 4 + fg;
                        *reinterpret_cast<f32x4*>(ep + row * 256 + ((c16 ^ (row & 15)) * 16)) = acc[fn][hh * 4 + f4 < FM ? hh * 4 + f4 : 0];
                    }
                }
            }
            if (hh == 0) fetch_resid(1);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (j < nf * 4) {
                    const int row = j * 4 + rr;
                    f32x4 v = *reinterpret_cast<const f32x4*>(ep + row * 256 + ((rc ^ (row & 15)) * 16));
                    const int m = mrow0 + hh * 64 + row;
                    if (m < M && ncol < N) {
                        v += b4;
                        if (drop_t) {       // nn.Dropout on the Linear output, before the residual add (vit.py:24,48 + :80-81)
                            const unsigned hrow = drop_row((unsigned)m, drop_seed);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = drop_keep(hrow, (unsigned)(ncol + e), drop_t) ? v[e] * inv_keep : 0.f;
                        }
                        v += rpre[hh][j];
                        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(Cv) + (long long)m * ldc + ncol) = v;
                    }
                }
            }
        }
    } else {
        f32x4 b4[4];
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) {
            b4[fn] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (EPI == VITK_EPI_BIAS || EPI == VITK_EPI_BIAS_GELU) {
                const int n = ncol0 + fn * 16 + 4 * fg;
                if (n < N) b4[fn] = load4<__bf16>(bias + n);
            }
        }
        const int rr = lane >> 3, rc = lane & 7;       // read-back: 8 rows x 8 chunks of 8 bf16
        const int ncol = ncol0 + rc * 8;
        bf16x8 hpre[2 * FM];                           // GELU_BWD: the saved pre-activations, fetched before the staging
        if constexpr (EPI == VITK_EPI_GELU_BWD) {
#pragma unroll
            for (int j = 0; j < 2 * FM; ++j) {
                const int m = mrow0 + j * 8 + rr;
                hpre[j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                if (m < M && ncol < N) hpre[j] = *reinterpret_cast<const bf16x8*>(aux + (long long)m * ldc + ncol);
            }
        }
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
            const int row = fm * 16 + fi;
#pragma unroll
            for (int fn = 0; fn < 4; ++fn) {
                const int c16 = fn * 2 + (fg >> 1);
                const f32x4 v = acc[fn][fm] + b4[fn];
                const bf16x4 pk = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                // rows r and r + 8 share a 16-byte slot column after the XOR: they take opposite 8-byte halves of it, so the 16
                // lanes of a ds_write_b64 group cover 32 distinct banks (was a 2-way conflict, ~2 % of the kernel by PMC)
                *reinterpret_cast<bf16x4*>(ep + row * 128 + ((c16 ^ (row & 7)) * 16) + (((fg & 1) ^ ((row >> 3) & 1)) * 8)) = pk;
            }
        }
        float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // GELU_BWD: column sums of this lane's rows (bias gradient)
        float f8max = 0.f;                                        // BIAS_GELU with an fp8 side output: max |activation|
#pragma unroll
        for (int j = 0; j < 2 * FM; ++j) {
            const int row = j * 8 + rr;
            bf16x8 v = *reinterpret_cast<const bf16x8*>(ep + row * 128 + ((rc ^ (row & 7)) * 16));
            if (j & 1) v = bf16x8{v[4], v[5], v[6], v[7], v[0], v[1], v[2], v[3]};     // rows 8..15 of a 16-row group: halves swapped
            const int m = mrow0 + row;
            if (m < M && ncol < N) {
                const long long o = (long long)m * ldc + ncol;
                if constexpr (EPI == VITK_EPI_NONE || EPI == VITK_EPI_BIAS) {
                    *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(Cv) + o) = v;
                } else if constexpr (EPI == VITK_EPI_BIAS_GELU) {
                    *reinterpret_cast<bf16x8*>(aux + o) = v;
                    bf16x8 g8;
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const f32x2 g2 = gelu_fast2(f32x2{(float)v[e], (float)v[e + 1]});
                        g8[e] = (__bf16)g2[0]; g8[e + 1] = (__bf16)g2[1];
                    }
                    if (drop_t) {           // nn.Dropout after the GELU (vit.py:22): the saved pre-activation stays undropped
                        const unsigned hrow = drop_row((unsigned)m, drop_seed);
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            g8[e] = drop_keep(hrow, (unsigned)(ncol + e), drop_t) ? (__bf16)((float)g8[e] * inv_keep) : (__bf16)0.f;
                    }
                    if (Cv) *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(Cv) + o) = g8;      // (null: only the e4m3 copy is wanted)
                    if (f8.p || f8.amax) {        // e4m3 copy of the activation for the next GEMM (+ this step's amax)
                        const f32x4 lo = {(float)g8[0], (float)g8[1], (float)g8[2], (float)g8[3]};
                        const f32x4 hi = {(float)g8[4], (float)g8[5], (float)g8[6], (float)g8[7]};
                        if (f8.p) {
                            const float sc8 = *f8.scale;
                            unsigned* d8 = reinterpret_cast<unsigned*>(f8.p + o);
                            d8[0] = pack_fp8x4(lo, sc8); d8[1] = pack_fp8x4(hi, sc8);
                        }
                        if (f8.amax) f8max = fmaxf(f8max, fmaxf(absmax4(lo), absmax4(hi)));
                    }
                } else if constexpr (EPI == VITK_EPI_GELU_BWD) {
                    const bf16x8 h8 = hpre[j];
                    bf16x8 g8;
                    float km[8];            // dropout factor of the forward's dropout(gelu(pre)) at (m, n): same decision, same 1 / (1 - p)
                    if (drop_t) {
                        const unsigned hrow = drop_row((unsigned)m, drop_seed);
#pragma unroll
                        for (int e = 0; e < 8; ++e) km[e] = drop_keep(hrow, (unsigned)(ncol + e), drop_t) ? inv_keep : 0.f;
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) km[e] = 1.f;
                    }
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const f32x2 g2 = f32x2{(float)v[e], (float)v[e + 1]} * gelu_grad_fast2(f32x2{(float)h8[e], (float)h8[e + 1]}) * f32x2{km[e], km[e + 1]};
                        g8[e] = (__bf16)g2[0]; g8[e + 1] = (__bf16)g2[1];
                        cs[e] += (float)g8[e]; cs[e + 1] += (float)g8[e + 1];      // of the ROUNDED values: what a later colsum(C) would read
                    }
                    *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(Cv) + o) = g8;
                    if (f8.p || f8.amax) {        // e5m2 copy of this gradient for the two GEMMs that read it (+ this step's amax)
                        const f32x4 lo = {(float)g8[0], (float)g8[1], (float)g8[2], (float)g8[3]};
                        const f32x4 hi = {(float)g8[4], (float)g8[5], (float)g8[6], (float)g8[7]};
                        if (f8.p) {
                            const float sc8 = *f8.scale;
                            unsigned* d8 = reinterpret_cast<unsigned*>(f8.p + o);
                            d8[0] = pack_bf8x4(lo, sc8); d8[1] = pack_bf8x4(hi, sc8);
                        }
                        if (f8.amax) f8max = fmaxf(f8max, fmaxf(absmax4(lo), absmax4(hi)));
                    }
                }
            }
        }
        if constexpr (EPI == VITK_EPI_BIAS_GELU || EPI == VITK_EPI_GELU_BWD) {
            if (f8.amax) {     // one atomic per workgroup, through the first words of its (idle) LDS
                f8max = wave_max(f8max);
                __syncthreads();
                float* red = reinterpret_cast<float*>(lds);
                if (lane == 0) red[wave] = f8max;
                __syncthreads();
                if (tid == 0) {
                    float m = red[0];
#pragma unroll
                    for (int k = 1; k < 8; ++k) m = fmaxf(m, red[k]);
                    atomicMax(f8.amax + (blockIdx.x & 63), __builtin_bit_cast(unsigned, m));
                }
            }
        }
        if constexpr (EPI == VITK_EPI_GELU_BWD) {
            // db = colsum(C) comes for free: the 8 row-groups of a wave are combined through its (now idle) LDS slice and
            // every wave writes one partial row of 64 sums; vitk_colsum_partials() folds the 2 * tiles_m rows.
            if (csum) {
                float* cp = reinterpret_cast<float*>(ep);
#pragma unroll
                for (int e = 0; e < 8; ++e) cp[rr * 64 + rc * 8 + e] = cs[e];
                float t = 0.f;
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8) t += cp[r8 * 64 + lane];
                const int n = ncol0 + lane;
                if (n < N) csum[(long long)(tm * 2 + wm) * N + n] = t;
            }
        }
    }
}

template <typename Kern>
int set_max_lds(Kern kernel, int bytes) {
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

}  // namespace

namespace {
struct NtPlan { bool large; int fm; };
NtPlan nt_plan(int64_t M, int64_t N, int64_t K, int64_t ldc, const void* aux) {
    NtPlan pl;
    pl.large = (K % L_BK == 0) && M >= 1024 && N >= 256 && (N % 8 == 0) && (ldc % 8 == 0) && (!aux || aligned16(aux));
    // Tile height: 256 rows (8 m-fragments per wave) or 224 (7).  One workgroup per CU, so the grid is
    // quantised in rounds of 256 tiles: pick the height whose (rounds x height) is smaller -- e.g. M = 50,432,
    // N = 768: 591 tiles of 256 rows need 3 rounds for 2.31 rounds of work, 678 tiles of 224 rows need 3
    // rounds of 7/8-size tiles (-12.5 %).
    pl.fm = 8;
    if (pl.large) {
        const long long tn_ = (N + L_BN - 1) / L_BN;
        const long long t8 = ((M + 255) / 256) * tn_, t7 = ((M + 223) / 224) * tn_;
        const long long cost8 = ((t8 + 255) / 256) * 8, cost7 = ((t7 + 255) / 256) * 7;
        if (cost7 * 10 <= cost8 * 9) pl.fm = 7;   // only when it buys >= 10 %: the shorter tile re-uses each W fragment 7x instead of 8x
    }
    return pl;
}
int gemm_nt_impl(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                 int epilogue, const void* bias, const float* resid, void* aux, float* csum, void* stream, bool fp8 = false,
                 float alpha = 1.0f, float drop_p = 0.f, unsigned drop_seed = 0u, const float* alpha_a = nullptr,
                 const float* alpha_w = nullptr, F8Out f8 = F8Out{nullptr, nullptr, nullptr}, int f8kind = 0);
}  // namespace

extern "C" int vitk_gemm_nt_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M,
                                 int64_t N, int64_t K, int epilogue, const void* bias, const float* resid, void* aux,
                                 void* stream) {
    return gemm_nt_impl(A, lda, W, ldw, C, ldc, M, N, K, epilogue, bias, resid, aux, nullptr, stream);
}

extern "C" int vitk_gemm_nt_bf16_drop(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M, int64_t N,
                                      int64_t K, int epilogue, const void* bias, const float* resid, void* aux, float* colsum_partials,
                                      float drop_p, uint32_t drop_seed, void* stream) {
    if (colsum_partials && (epilogue != VITK_EPI_GELU_BWD || vitk_gemm_nt_colsum_rows(M, N, K, ldc) == 0))
        VITK_FAIL(VITK_E_SHAPE, "gemm_nt_bf16_drop: column sums come with the GELU_BWD epilogue of the 256-row kernel only");
    return gemm_nt_impl(A, lda, W, ldw, C, ldc, M, N, K, epilogue, bias, resid, aux, colsum_partials, stream, false, 1.0f, drop_p, drop_seed);
}

extern "C" int vitk_gemm_nt_fp8_ex(const void* A, int64_t lda, int a_is_fp8, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M,
                                   int64_t N, int64_t K, int epilogue, const void* bias, const float* resid, void* aux, float alpha,
                                   const float* alpha_a, const float* alpha_w, void* c8, const float* c8_scale, uint32_t* c8_amax64,
                                   void* stream) {
    const F8Out f8{(unsigned char*)c8, c8_scale, (unsigned*)c8_amax64};
    return gemm_nt_impl(A, lda, W, ldw, C, ldc, M, N, K, epilogue, bias, resid, aux, nullptr, stream, a_is_fp8 != 0, alpha, 0.f, 0u,
                        alpha_a, alpha_w, f8);
}

extern "C" int vitk_gemm_nt_fp8(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M, int64_t N,
                               int64_t K, int epilogue, const void* bias, const float* resid, void* aux, float alpha, void* stream) {
    return gemm_nt_impl(A, lda, W, ldw, C, ldc, M, N, K, epilogue, bias, resid, aux, nullptr, stream, true, alpha);
}

// a_kind: 0 = A in the 16-bit type (recording pass), 1 = e4m3, 2 = e5m2 (gradients; W stays e4m3).  flags bit 0: the K = 128 MFMA.
extern "C" int vitk_gemm_nt_fp8_v2(const void* A, int64_t lda, int a_kind, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M,
                                   int64_t N, int64_t K, int epilogue, const void* bias, const float* resid, void* aux,
                                   float* colsum_partials, float alpha, const float* alpha_a, const float* alpha_w, void* c8,
                                   const float* c8_scale, uint32_t* c8_amax64, int flags, void* stream) {
    if (a_kind < 0 || a_kind > 2) VITK_FAIL(VITK_E_ARG, "gemm_nt_fp8_v2: a_kind is 0 (16-bit), 1 (e4m3) or 2 (e5m2), got %d", a_kind);
    if (flags & ~1) VITK_FAIL(VITK_E_ARG, "gemm_nt_fp8_v2: unknown flags 0x%x", (unsigned)flags);
    if (a_kind == 0 && (flags & 1)) VITK_FAIL(VITK_E_ARG, "gemm_nt_fp8_v2: the K = 128 instruction takes 1-byte operands");
    if ((flags & 1) && (K % 128)) VITK_FAIL(VITK_E_SHAPE, "gemm_nt_fp8_v2: the K = 128 flavour needs K %% 128 == 0 (K = %lld)", (long long)K);
    if (colsum_partials && (epilogue != VITK_EPI_GELU_BWD || vitk_gemm_nt_fp8_colsum_rows(M, N, K, ldc) == 0))
        VITK_FAIL(VITK_E_SHAPE, "gemm_nt_fp8_v2: column sums come with the GELU_BWD epilogue of the 256-row kernel only");
    const F8Out f8{(unsigned char*)c8, c8_scale, (unsigned*)c8_amax64};
    return gemm_nt_impl(A, lda, W, ldw, C, ldc, M, N, K, epilogue, bias, resid, aux, colsum_partials, stream, a_kind != 0, alpha, 0.f, 0u,
                        alpha_a, alpha_w, f8, (a_kind == 2 ? 1 : 0) | ((flags & 1) ? 2 : 0));
}

// Partial rows the GELU_BWD epilogue of the fp8 (per-tile 256-row) kernel writes: two per row tile; 0 = shape not served.
extern "C" int64_t vitk_gemm_nt_fp8_colsum_rows(int64_t M, int64_t N, int64_t K, int64_t ldc) {
    if (M <= 0 || N <= 0 || K <= 0 || (K % 64)) return 0;
    const NtPlan pl = nt_plan(M, N, K, ldc, nullptr);
    if (!pl.large) return 0;
    return 2 * ((M + 32 * pl.fm - 1) / (32 * pl.fm));
}

static int64_t nt_persist_colsum_rows(const NtpPlan& q, int64_t M, int64_t N, int64_t K, int64_t ldc);
extern "C" int64_t vitk_gemm_nt_colsum_rows(int64_t M, int64_t N, int64_t K, int64_t ldc) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const NtpPlan q = ntp_plan(M, N, K, ldc, nullptr);      // the persistent kernel serves every 16-bit call it accepts
    if (q.ok) return nt_persist_colsum_rows(q, M, N, K, ldc);
    const NtPlan pl = nt_plan(M, N, K, ldc, nullptr);
    if (!pl.large) return 0;
    return 2 * ((M + 32 * pl.fm - 1) / (32 * pl.fm));
}

extern "C" int vitk_gemm_nt_plan(int64_t M, int64_t N, int64_t K, int64_t ldc, int32_t* out5) {
    if (!out5) VITK_FAIL(VITK_E_ARG, "gemm_nt_plan: null output");
    for (int i = 0; i < 5; ++i) out5[i] = 0;
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const NtpPlan q = ntp_plan(M, N, K, ldc, nullptr);
    if (q.ok) { out5[0] = 1; out5[1] = q.tm_main; out5[2] = q.tail_tm; out5[3] = q.grid; out5[4] = q.tiles_n; }
    return 0;
}

extern "C" int vitk_gemm_nt_bf16_gelu_bwd_colsum(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                                                 int64_t M, int64_t N, int64_t K, void* aux, float* colsum_partials, void* stream) {
    if (!colsum_partials) VITK_FAIL(VITK_E_ARG, "gemm_nt_bf16_gelu_bwd_colsum: null partials");
    if (vitk_gemm_nt_colsum_rows(M, N, K, ldc) == 0 || (aux && !aligned16(aux)))
        VITK_FAIL(VITK_E_SHAPE, "gemm_nt_bf16_gelu_bwd_colsum: shape not served by the 256-row kernel (vitk_gemm_nt_colsum_rows() == 0)");
    return gemm_nt_impl(A, lda, W, ldw, C, ldc, M, N, K, VITK_EPI_GELU_BWD, nullptr, nullptr, aux, colsum_partials, stream);
}

extern "C" int vitk_gemm_nt_bf16_mul_aux_colsum(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M,
                                                int64_t N, int64_t K, const void* aux, float* colsum_partials, void* stream) {
    if (!aux) VITK_FAIL(VITK_E_ARG, "gemm_nt_bf16_mul_aux_colsum: null pointer");
    if (colsum_partials && vitk_gemm_nt_colsum_rows(M, N, K, ldc) == 0)
        VITK_FAIL(VITK_E_SHAPE, "gemm_nt_bf16_mul_aux_colsum: shape not served by the persistent kernel (vitk_gemm_nt_colsum_rows() == 0)");
    return gemm_nt_impl(A, lda, W, ldw, C, ldc, M, N, K, VITK_EPI_MUL_AUX, nullptr, nullptr, const_cast<void*>(aux), colsum_partials, stream);
}

extern "C" int vitk_gemm_nt_bf16_mul_aux8_colsum(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M,
                                                 int64_t N, int64_t K, const void* aux8, float* colsum_partials, void* stream) {
    if (!aux8) VITK_FAIL(VITK_E_ARG, "gemm_nt_bf16_mul_aux8_colsum: null pointer");
    if (colsum_partials && vitk_gemm_nt_colsum_rows(M, N, K, ldc) == 0)
        VITK_FAIL(VITK_E_SHAPE, "gemm_nt_bf16_mul_aux8_colsum: shape not served by the persistent kernel (vitk_gemm_nt_colsum_rows() == 0)");
    return gemm_nt_impl(A, lda, W, ldw, C, ldc, M, N, K, VITK_EPI_MUL_AUX8, nullptr, nullptr, const_cast<void*>(aux8), colsum_partials, stream);
}

// ---- the persistent NT kernels: four-wave kernel (gemm_nt_w128.hip) on the full 256-row tiles of whole rounds, 8-wave kernel
// (gemm_nt_persist.hip) on the remaining rows (the partial last m-tile and what would be a mostly idle last round, as 128-row tiles) ----
// The row split is a function of (M, N, K) and the CU count only (and of the test switch VITK_NT_W128): vitk_gemm_nt_colsum_rows() reports the
// partial rows of both launches, and the layout of those rows does NOT depend on vitk_set_cu_reserve() -- a caller that asks for the row
// count, allocates, and launches while another thread changes the reserve still gets the rows it was promised.  While another
// kernel is expected on the chip (cu_reserve = c > 0: the in-backward all-reduce) the four-wave kernel keeps its rows and is launched on
// 256 - c workgroups (round 6; its per-XCD static tile lists take any multiple of 8: more tiles per workgroup, none waiting for a CU the
// collective holds), and the 8-wave kernel draws its tiles by dynamic tickets as before.
// VITK_NT_W128 = bit mask over VITK_EPI_* of the epilogues the four-wave kernel serves (0 = none: the 8-wave kernel alone; unset = NTW_EPIS)
constexpr unsigned NTW_EPIS = 0xffu;
static int ntw_tiles_m(const NtpPlan& q, int64_t M, int64_t N, int64_t K, int epilogue, int grid = 0) {
    // (the 8-bit-factor pair follows the switches of its 16-bit twins)
    if (epilogue == VITK_EPI_BIAS_GELU_DG8) epilogue = VITK_EPI_BIAS_GELU_DG;
    if (epilogue == VITK_EPI_MUL_AUX8) epilogue = VITK_EPI_MUL_AUX;
    const char* e = vitk_switch("VITK_NT_W128");
    const unsigned mask = e ? (unsigned)strtoul(e, nullptr, 0) : NTW_EPIS;
    if (!q.ok || !((mask >> epilogue) & 1u) || !gemm_ntw_serves(M, N, K)) return 0;
    return gemm_ntw_split(M, N, K, grid > 0 ? grid : q.grid);
}
static int64_t nt_persist_colsum_rows(const NtpPlan& q, int64_t M, int64_t N, int64_t K, int64_t ldc) {
    // (the two epilogues with column sums, GELU_BWD and MUL_AUX, are switched together: one row count serves both entry points)
    const int tmw = ntw_tiles_m(q, M, N, K, VITK_EPI_MUL_AUX);
    if (tmw == 0) return ntp_colsum_rows(q);
    const int64_t rest = M - 256LL * tmw;
    return 2LL * tmw + (rest > 0 ? ntp_colsum_rows(ntp_plan(rest, N, K, ldc, nullptr)) : 0);
}
static int nt_persist_dispatch(const NtpPlan& q, const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M, int64_t N,
                               int64_t K, int epilogue, const void* bias, const float* resid, void* aux, float* csum, unsigned drop_t,
                               unsigned drop_seed, float inv_keep, void* stream) {
    // the workgroups of the four-wave launch: one per CU that is expected to be free (a multiple of 8, at least a quarter of the chip)
    const int reserve = vitk_get_cu_reserve();
    int grid_w = q.grid - (reserve + 7) / 8 * 8;
    if (grid_w < q.grid / 4) grid_w = q.grid / 4 / 8 * 8;
    // the row split: with column-sum partial rows promised (csum) it is the full-chip one whatever the reserve; otherwise it follows the launch
    // grid (whole rounds of 256 - c workgroups: at N = 768 the full-chip split would leave a third round two thirds empty)
    const int tmw = ntw_tiles_m(q, M, N, K, epilogue == VITK_EPI_GELU_BWD ? VITK_EPI_MUL_AUX : epilogue, csum ? 0 : grid_w);
    const int64_t asz = (epilogue == VITK_EPI_BIAS_GELU_DG8 || epilogue == VITK_EPI_MUL_AUX8) ? 1 : 2;      // bytes per element of aux
    // fused dropout lives in the 8-wave kernel; only the epilogue with column sums has to keep the row split (its partial rows are promised)
    if (tmw == 0 || (drop_t && !(epilogue == VITK_EPI_GELU_BWD && csum)))
        return gemm_ntp_launch(q, A, lda, W, ldw, C, ldc, M, N, K, epilogue, bias, resid, aux, csum, drop_t, drop_seed, inv_keep, stream);
    const int64_t rows_w = 256LL * tmw, rest = M - rows_w;
    int rc;
    if (drop_t) {       // the first rows as 256-row tiles of the 8-wave kernel (same partial rows as the four-wave launch would write)
        NtpPlan q1 = q;
        q1.tm_main = tmw; q1.tail_tm = 0; q1.n_main = tmw * q1.tiles_n; q1.n_tail = 0;
        rc = gemm_ntp_launch(q1, A, lda, W, ldw, C, ldc, rows_w, N, K, epilogue, bias, resid, aux, csum, drop_t, drop_seed, inv_keep, stream);
    } else {
        if (256LL * (lda > ldw ? lda : ldw) * 2 + K * 2 >= (1LL << 31)) VITK_FAIL(VITK_E_SHAPE, "gemm_nt_bf16: row stride too large (lda %lld, ldw %lld)", (long long)lda, (long long)ldw);
        rc = gemm_ntw_launch(tmw, grid_w, A, lda, W, ldw, C, ldc, N, K, epilogue, bias, resid, aux, csum, 0, 0, stream);
    }
    if (rc != 0 || rest <= 0) return rc;
    const NtpPlan q2 = ntp_plan(rest, N, K, ldc, aux);
    if (!q2.ok) VITK_FAIL(VITK_E_SHAPE, "gemm_nt_bf16: internal: the row split left %lld rows the persistent kernel does not take", (long long)rest);
    const int64_t csz = epilogue == VITK_EPI_RESID ? 4 : 2;
    return gemm_ntp_launch(q2, (const char*)A + rows_w * lda * 2, lda, W, ldw, (char*)C + rows_w * ldc * csz, ldc, rest, N, K, epilogue, bias,
                           resid ? (const float*)((const char*)resid + rows_w * ldc * csz) : nullptr, aux ? (char*)aux + rows_w * ldc * asz : nullptr,
                           csum ? csum + 2LL * tmw * N : nullptr, drop_t, drop_seed, inv_keep, stream, (unsigned)rows_w);
}

namespace {
int gemm_nt_impl(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                 int epilogue, const void* bias, const float* resid, void* aux, float* csum, void* stream, bool fp8, float alpha,
                 float drop_p, unsigned drop_seed, const float* alpha_a, const float* alpha_w, F8Out f8, int f8kind) {
    if (f8kind && !fp8) VITK_FAIL(VITK_E_ARG, "gemm_nt: e5m2 / K = 128 flavours need 1-byte operands");
    if ((f8kind & 1) && epilogue != VITK_EPI_NONE && epilogue != VITK_EPI_GELU_BWD)
        VITK_FAIL(VITK_E_ARG, "gemm_nt_fp8: e5m2 operands (gradients) come with the NONE and GELU_BWD epilogues only (got %d)", epilogue);
    if ((f8.p || f8.amax) && epilogue != VITK_EPI_BIAS_GELU && epilogue != VITK_EPI_GELU_BWD)
        VITK_FAIL(VITK_E_ARG, "gemm_nt: the fp8 side output exists in the BIAS_GELU (e4m3) and GELU_BWD (e5m2) epilogues only");
    if (f8.p && (!f8.scale || (ldc & 7))) VITK_FAIL(VITK_E_ARG, "gemm_nt: fp8 side output needs a scale and ldc %% 8 == 0");
    // C may be null where the 16-bit output is not wanted: fp8 operands, BIAS_GELU, with the e4m3 copy as the product
    const bool c_optional = fp8 && epilogue == VITK_EPI_BIAS_GELU && f8.p;
    if (!A || !W || (!C && !c_optional)) VITK_FAIL(VITK_E_ARG, "gemm_nt_bf16: null pointer");
    if (!(drop_p >= 0.f && drop_p < 1.f)) VITK_FAIL(VITK_E_ARG, "gemm_nt_bf16: dropout p must be in [0, 1) (got %g)", (double)drop_p);
    const unsigned drop_t = drop_thresh(drop_p);
    const float inv_keep = 1.0f / (1.0f - drop_p);
    if (fp8 && ((lda & 15) || (ldw & 15) || (K % 64)))
        VITK_FAIL(VITK_E_ALIGN, "gemm_nt_fp8: K %% 64 == 0 and lda, ldw %% 16 == 0 required");
    if (M <= 0 || N <= 0 || K <= 0 || (K % NT_BK) || (N & 3) || M > (1 << 30) || N > (1 << 30))
        VITK_FAIL(VITK_E_SHAPE, "gemm_nt_bf16: need K %% 32 == 0 and N %% 4 == 0 (M=%lld N=%lld K=%lld)", (long long)M, (long long)N, (long long)K);
    if ((lda & 7) || (ldw & 7) || (ldc & 3) || !aligned16(A) || !aligned16(W) || !aligned16(C) || (bias && !aligned8(bias)) ||
        (resid && !aligned16(resid)) || (aux && !aligned8(aux)))
        VITK_FAIL(VITK_E_ALIGN, "gemm_nt_bf16: lda/ldw %% 8, ldc %% 4 and 16-byte aligned pointers required");
    if (ldw == 0 && (fp8 || f8.p || f8.amax)) VITK_FAIL(VITK_E_ARG, "gemm_nt: a K-blocked W (ldw == 0) goes with 16-bit operands only");
    if (!fp8 && !f8.p && !f8.amax) {
        // 16-bit operands at M >= 1024: the persistent kernel (gemm_nt_persist.hip)
        // Every epilogue goes to the persistent kernel.  [measured] kernel by kernel (tools/nt_ab.py, one stream) it wins where the
        // epilogue is heavy (f32 residual 1.04-1.12x, bias+GELU 1.07x) and is level or slightly behind on plain 16-bit stores
        // (0.94-1.0x); inside the training step, with the weight-gradient GEMMs running beside it on the side stream, all five on
        // the persistent kernel is the fastest setting: 40.17 ms/step vs 40.95 (plain stores per-tile) vs 41.60 (all per-tile).
        // VITK_NTP_EPIS = bit mask over VITK_EPI_* overrides (GELU_BWD always: its column-sum rows follow the persistent plan).
        const NtpPlan q = ntp_plan(M, N, K, ldc, aux);
        const unsigned epis = vitk_exp("VITK_NTP_EPIS") ? (unsigned)atoi(vitk_exp("VITK_NTP_EPIS")) : 0x1fu;
        const bool dg_out = epilogue == VITK_EPI_BIAS_GELU_DG || epilogue == VITK_EPI_BIAS_GELU_DG8;
        if (dg_out || epilogue == VITK_EPI_MUL_AUX || epilogue == VITK_EPI_MUL_AUX8) {       // the gelu'-factor pairs (round 4; 8-bit factor: round 5): the persistent kernel only
            if (!q.ok) VITK_FAIL(VITK_E_SHAPE, "gemm_nt_bf16: EPI_BIAS_GELU_DG(8) / EPI_MUL_AUX(8) are served by the persistent kernel only (vitk_gemm_nt_plan() says which shapes)");
            if (!aux || drop_t || (dg_out && !bias)) VITK_FAIL(VITK_E_ARG, "gemm_nt_bf16: EPI_BIAS_GELU_DG(8) needs bias and aux, EPI_MUL_AUX(8) aux; no fused dropout");
            return nt_persist_dispatch(q, A, lda, W, ldw, C, ldc, M, N, K, epilogue, bias, resid, aux, csum, 0u, 0u, 1.0f, stream);
        }
        if (epilogue == VITK_EPI_RESID16) {       // 16-bit forward residual stream (opt-in): the persistent kernel only
            if (!q.ok) VITK_FAIL(VITK_E_SHAPE, "gemm_nt_bf16: EPI_RESID16 is served by the persistent kernel only (vitk_gemm_nt_plan() says which shapes)");
            if (!resid || !aligned8(resid) || drop_t) VITK_FAIL(VITK_E_ARG, "gemm_nt_bf16: EPI_RESID16 needs an 8-byte aligned 16-bit resid and no dropout");
            return nt_persist_dispatch(q, A, lda, W, ldw, C, ldc, M, N, K, epilogue, bias, resid, aux, csum, 0u, 0u, 1.0f, stream);
        }
        if (q.ok && epilogue >= 0 && epilogue <= 4 && (((epis >> epilogue) & 1u) || epilogue == VITK_EPI_GELU_BWD)) {
            switch (epilogue) {
                case VITK_EPI_NONE: break;
                case VITK_EPI_BIAS: if (!bias) VITK_FAIL(VITK_E_ARG, "gemm_nt_bf16: EPI_BIAS needs bias"); break;
                case VITK_EPI_BIAS_GELU: if (!bias || !aux) VITK_FAIL(VITK_E_ARG, "gemm_nt_bf16: EPI_BIAS_GELU needs bias and aux"); break;
                case VITK_EPI_RESID: if (!resid) VITK_FAIL(VITK_E_ARG, "gemm_nt_bf16: EPI_RESID needs resid"); break;
                case VITK_EPI_GELU_BWD: if (!aux) VITK_FAIL(VITK_E_ARG, "gemm_nt_bf16: EPI_GELU_BWD needs aux"); break;
                default: VITK_FAIL(VITK_E_ARG, "gemm_nt_bf16: bad epilogue %d", epilogue);
            }
            if (drop_t && epilogue != VITK_EPI_RESID && epilogue != VITK_EPI_BIAS_GELU && epilogue != VITK_EPI_GELU_BWD)
                VITK_FAIL(VITK_E_SHAPE, "gemm_nt_bf16: fused dropout exists for the RESID / BIAS_GELU / GELU_BWD epilogues only");
            return nt_persist_dispatch(q, A, lda, W, ldw, C, ldc, M, N, K, epilogue, bias, resid, aux, csum, drop_t, drop_seed, inv_keep, stream);
        }
    }
    if (ldw == 0) VITK_FAIL(VITK_E_SHAPE, "gemm_nt_bf16: a K-blocked W (ldw == 0, vitk_pack_w_nt) is read by the persistent kernel only (vitk_gemm_nt_plan() says which shapes it serves)");
    const NtPlan pl = nt_plan(M, N, K, ldc, aux);
    const bool large = pl.large;
    const int fm = pl.fm;
    if (fp8 && !large) VITK_FAIL(VITK_E_SHAPE, "gemm_nt_fp8: served by the 256-row kernel only (M >= 1024, N >= 256, N %% 8 == 0)");
    if ((f8.p || f8.amax) && !large) VITK_FAIL(VITK_E_SHAPE, "gemm_nt: the fp8 side output is served by the 256-row kernel only");
    if (drop_t && (!large || (epilogue != VITK_EPI_RESID && epilogue != VITK_EPI_BIAS_GELU && epilogue != VITK_EPI_GELU_BWD)))
        VITK_FAIL(VITK_E_SHAPE, "gemm_nt_bf16: fused dropout exists in the 256-row kernel for the RESID / BIAS_GELU / GELU_BWD epilogues only");
    const int tbm = large ? 32 * fm : BM, tbn = large ? L_BN : BN;
    const int tiles_m = (int)((M + tbm - 1) / tbm), tiles_n = (int)((N + tbn - 1) / tbn);
    const long long nwg = (long long)tiles_m * tiles_n;
    if (nwg > 0x7fffffffLL) VITK_FAIL(VITK_E_SHAPE, "gemm_nt_bf16: grid too large");
    hipStream_t st = (hipStream_t)stream;
    int group_n = tiles_n;                                  // n-tiles per group of the grouped tile order
    if (large && tiles_n > 8) group_n = (tiles_n + (tiles_n + 5) / 6 - 1) / ((tiles_n + 5) / 6);
    if (vitk_exp("VITK_GROUP_N")) group_n = atoi(vitk_exp("VITK_GROUP_N")) > 0 ? atoi(vitk_exp("VITK_GROUP_N")) : tiles_n;
    if (group_n > tiles_n) group_n = tiles_n;
#define NT_LAUNCH_F8(E, F, FK) do { \
            static const int rc8__ = set_max_lds(gemm_nt256pp_kernel<E, F, 1, FK>, P_LDS_BYTES); \
            if (rc8__ != 0) VITK_FAIL(rc8__, "gemm_nt_fp8: cannot enable %d B of LDS", P_LDS_BYTES); \
            hipLaunchKernelGGL((gemm_nt256pp_kernel<E, F, 1, FK>), dim3((unsigned)nwg), dim3(512), P_LDS_BYTES, st, A, (long long)lda, \
                W, (long long)ldw, C, (long long)ldc, (int)M, (int)N, (int)K, (const __bf16*)bias, resid, (__bf16*)aux, tiles_n, (int)nwg, group_n, csum, alpha, drop_t, drop_seed, inv_keep, alpha_a, alpha_w, f8); \
        } while (0)
    // e5m2 flavours (f8kind bit 0) are instantiated for the two epilogues the backward uses (checked above)
#define NT_LAUNCH_PP(E, F) do { \
        if (fp8) { \
            if constexpr (E == VITK_EPI_NONE || E == VITK_EPI_GELU_BWD) { \
                if (f8kind == 1) NT_LAUNCH_F8(E, F, 1); else if (f8kind == 3) NT_LAUNCH_F8(E, F, 3); \
                else if (f8kind == 2) NT_LAUNCH_F8(E, F, 2); else NT_LAUNCH_F8(E, F, 0); \
            } else { \
                if (f8kind == 2) NT_LAUNCH_F8(E, F, 2); else NT_LAUNCH_F8(E, F, 0); \
            } \
        } else { \
            static const int rc__ = set_max_lds(gemm_nt256pp_kernel<E, F, 2>, P_LDS_BYTES); \
            if (rc__ != 0) VITK_FAIL(rc__, "gemm_nt_bf16: cannot enable %d B of LDS", P_LDS_BYTES); \
            hipLaunchKernelGGL((gemm_nt256pp_kernel<E, F, 2>), dim3((unsigned)nwg), dim3(512), P_LDS_BYTES, st, A, (long long)lda, \
                W, (long long)ldw, C, (long long)ldc, (int)M, (int)N, (int)K, (const __bf16*)bias, resid, (__bf16*)aux, tiles_n, (int)nwg, group_n, csum, 1.0f, drop_t, drop_seed, inv_keep, nullptr, nullptr, f8); \
        } \
    } while (0)
#define NT_LAUNCH(E) do { \
    if (large && fm == 7) NT_LAUNCH_PP(E, 7); \
    else if (large) NT_LAUNCH_PP(E, 8); \
    else { \
        hipLaunchKernelGGL((gemm_nt_kernel<E>), dim3((unsigned)nwg), dim3(256), 0, st, (const __bf16*)A, (long long)lda, \
            (const __bf16*)W, (long long)ldw, C, (long long)ldc, (int)M, (int)N, (int)K, (const __bf16*)bias, resid, (__bf16*)aux, tiles_n, (int)nwg); \
    } } while (0)
    switch (epilogue) {
        case VITK_EPI_NONE: NT_LAUNCH(VITK_EPI_NONE); break;
        case VITK_EPI_BIAS:
            if (!bias) VITK_FAIL(VITK_E_ARG, "gemm_nt_bf16: EPI_BIAS needs bias");
            NT_LAUNCH(VITK_EPI_BIAS); break;
        case VITK_EPI_BIAS_GELU:
            if (!bias || !aux) VITK_FAIL(VITK_E_ARG, "gemm_nt_bf16: EPI_BIAS_GELU needs bias and aux");
            NT_LAUNCH(VITK_EPI_BIAS_GELU); break;
        case VITK_EPI_RESID:
            if (!resid) VITK_FAIL(VITK_E_ARG, "gemm_nt_bf16: EPI_RESID needs resid");
            NT_LAUNCH(VITK_EPI_RESID); break;
        case VITK_EPI_GELU_BWD:
            if (!aux) VITK_FAIL(VITK_E_ARG, "gemm_nt_bf16: EPI_GELU_BWD needs aux");
            NT_LAUNCH(VITK_EPI_GELU_BWD); break;
        default: VITK_FAIL(VITK_E_ARG, "gemm_nt_bf16: bad epilogue %d", epilogue);
    }
#undef NT_LAUNCH
#undef NT_LAUNCH_PP
#undef NT_LAUNCH_F8
    VITK_CHECK_LAUNCH("gemm_nt_bf16");
    return 0;
}
}  // namespace


bool gemm_tn_w128_serves(int64_t M, int64_t N, int64_t K, int64_t ldy, int64_t ldx, int64_t splits);
int gemm_tn_w128_launch2(const void* dY0, int64_t ldy0, const void* X0, int64_t ldx0, int64_t N0, int64_t K0, const void* dY1, int64_t ldy1,
                         const void* X1, int64_t ldx1, int64_t N1, int64_t K1, float* ws, int64_t M, int64_t splits, void* stream);
int gemm_tn_w128_launch(const void* dY, int64_t ldy, const void* X, int64_t ldx, float* ws, int64_t M, int64_t N, int64_t K,
                        int64_t splits, void* stream);

static bool tn_large(int64_t M, int64_t N, int64_t K) { return M >= 4096 && N >= 256 && K >= 256; }

// CUs to leave to OTHER kernels (an RCCL collective overlapping the backward): the weight-gradient GEMM launches one (tile, M-split)
// job per workgroup and every job needs a whole CU -- with c CUs taken, c of ~250 jobs of a full-chip launch run as a second
// round (2x the launch time); planned for 256 - c CUs all jobs run at once beside the collective (1 / (1 - c / 256) x).
static std::atomic<int> g_cu_reserve{0};
extern "C" int vitk_set_cu_reserve(int cus) {
    if (cus < 0 || cus > 192) VITK_FAIL(VITK_E_ARG, "set_cu_reserve: 0 .. 192 (got %d)", cus);
    g_cu_reserve.store(cus);
    return 0;
}
extern "C" int vitk_get_cu_reserve(void) { return g_cu_reserve.load(); }

// M-splits of the four-wave weight-gradient kernel: every (256 x 256 tile, split) job is one workgroup that needs a whole CU, so the launch
// runs in ROUNDS of `cus` jobs.  Rounds 1-5 took "about one job per CU" (s = 256 / tiles); at ViT-H/14's widths that rule gives 100 tiles x 3
// = 300 jobs = two rounds, the second 17 % full (the weight gradients of `--config vit_h14 --precision bf16` ran at 0.38 of the MFMA peak
// against 0.50 at ViT-B/16: VERDICT r05).  Now the split minimises a three-term model: rounds x (rows per job x 24 ns + 8 us of prologue and
// slab store) + the fold's slab traffic (s f32 slabs written and read back at ~5 TB/s) [constants from the ViT-B/16 launches: 7,205 rows
// per job in 175-181 us].  ViT-B/16 (36 tiles: 7) and ViT-L/16 (64 tiles: 4) keep their splits; 100 tiles at M = 147,712 take 5 (500 jobs =
// 1.95 rounds) instead of 3.
int64_t tn_pick_splits(int64_t tiles, int64_t M, int cus, long long slab_elems, double us_per_row, int64_t min_rows) {
    const int64_t max_by_rows = (M + min_rows - 1) / min_rows;      // 16-bit kernel: at least 8 steps of 64 rows per split
    int64_t best = 1;
    double best_c = 1e30;
    for (int64_t s = 1; s <= 64 && s <= max_by_rows; ++s) {
        const int64_t rounds = (tiles * s + cus - 1) / cus;
        const double job = (double)M / (double)s * us_per_row + 8.0;            // us
        const double fold = (double)s * (double)slab_elems * 8.0 / 5.0e6;       // us
        const double c = (double)rounds * job + fold;
        if (c < best_c - 1e-9) { best_c = c; best = s; }
    }
    return best;
}

extern "C" int64_t vitk_gemm_tn_splits(int64_t M, int64_t N, int64_t K) {
    if (tn_large(M, N, K)) {
        const int64_t tiles = ((N + 255) / 256) * ((K + 255) / 256);
        return tn_pick_splits(tiles, M, 256 - g_cu_reserve.load(), (long long)N * K, 0.024, 512);
    }
    const int64_t tiles = ((N + BN - 1) / BN) * ((K + BM - 1) / BM);
    int64_t s = (768 + tiles - 1) / tiles;            // aim for ~3 blocks per CU
    const int64_t max_by_rows = (M + 255) / 256;      // at least 8 steps of 32 rows per split
    if (s > max_by_rows) s = max_by_rows;
    if (s > 64) s = 64;
    if (s < 1) s = 1;
    return s;
}

extern "C" int vitk_gemm_tn_bf16(const void* dY, int64_t ldy, const void* X, int64_t ldx, void* dW, int odt, int64_t ldo,
                                 int accumulate, int64_t M, int64_t N, int64_t K, float* ws, int64_t splits, void* stream) {
    if (!dY || !X || !dW || !ws) VITK_FAIL(VITK_E_ARG, "gemm_tn_bf16: null pointer");
    if (M <= 0 || N <= 0 || K <= 0 || (N & 7) || (K & 7) || splits < 1 || splits > 65535 || M > (1 << 30))
        VITK_FAIL(VITK_E_SHAPE, "gemm_tn_bf16: need N %% 8 == 0, K %% 8 == 0 (M=%lld N=%lld K=%lld)", (long long)M, (long long)N, (long long)K);
    if ((ldy & 7) || (ldx & 7) || (ldo & 3) || !aligned16(dY) || !aligned16(X) || !aligned16(dW) || !aligned16(ws))
        VITK_FAIL(VITK_E_ALIGN, "gemm_tn_bf16: ldy/ldx %% 8, ldo %% 4 and 16-byte aligned pointers required");
    hipStream_t st = (hipStream_t)stream;
    // gemm_tn_w128.hip (round 4: four waves, 128 x 128 wave tiles, pinned asm MFMAs, double-buffered fragments) serves the large shapes;
    // the two 8-wave kernels it replaced (x1.3-1.5 slower: gemm_tn256_kernel, register-staged, rounds 1-3, and gemm_tn_dma.hip) left the
    // tree in round 5.  Everything else -- small extents, splits whose 32-bit descriptor offsets would overflow -- runs the 128 x 128 kernel.
    if (tn_large(M, N, K) && gemm_tn_w128_serves(M, N, K, ldy, ldx, splits)) {
        const int rc = gemm_tn_w128_launch(dY, ldy, X, ldx, ws, M, N, K, splits, stream);
        if (rc != 0) return rc;
    } else {
        const int tiles_n = (int)((N + BN - 1) / BN), tiles_k = (int)((K + BM - 1) / BM);
        const int nwg = tiles_n * tiles_k;
        long long rps = (M + splits - 1) / splits;
        rps = (rps + TN_BKM - 1) / TN_BKM * TN_BKM;
        hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)(nwg * splits)), dim3(256), 0, st, (const __bf16*)dY, (long long)ldy,
                           (const __bf16*)X, (long long)ldx, ws, (int)M, (int)N, (int)K, (int)rps, tiles_k, nwg);
    }
    VITK_CHECK_LAUNCH("gemm_tn_bf16");
    const long long NK = (long long)N * K;
    const unsigned blocks = (unsigned)((NK / 4 + 255) / 256);
    VITK_DISPATCH_DT(odt, OT, hipLaunchKernelGGL((tn_reduce_kernel<OT>), dim3(blocks), dim3(256), 0, st, ws, (int)splits, NK, (int)K,
                                                  (OT*)dW, (long long)ldo, accumulate, NK));
    VITK_CHECK_LAUNCH("gemm_tn_reduce");
    return 0;
}

// ---- two weight gradients over the same token rows in ONE launch (gemm_tn_w128.hip: TnProblem) -------------------------------------------
// splits for the pair (0: not served -- call vitk_gemm_tn_bf16 twice): both problems large, the kernel's 32-bit offsets hold
extern "C" int64_t vitk_gemm_tn_pair_splits(int64_t M, int64_t N0, int64_t K0, int64_t N1, int64_t K1) {
    if (!tn_large(M, N0, K0) || !tn_large(M, N1, K1) || (N0 & 7) || (K0 & 7) || (N1 & 7) || (K1 & 7)) return 0;
    // VITK_TN_PAIR=0 switches it off.  [measured, profiles/r04_tn_pair_ab.log, r04_dw_stream_ab.log] with the launches of a step serialized (the
    // default since round 4) the pair is worth 0.2-0.3 ms of the ViT-B/16 step; beside a side stream it LOSES 0.5 ms (the deferred gradient no
    // longer overlaps the attention backward) -- engine.TransformerFn pairs only when its side stream is off.
    if (vitk_switch("VITK_TN_PAIR") && atoi(vitk_switch("VITK_TN_PAIR")) == 0) return 0;
    const int64_t tiles = ((N0 + 255) / 256) * ((K0 + 255) / 256) + ((N1 + 255) / 256) * ((K1 + 255) / 256);
    const int64_t s = tn_pick_splits(tiles, M, 256 - g_cu_reserve.load(), (long long)N0 * K0 + (long long)N1 * K1, 0.024, 512);
    const int64_t ldmax = (N0 > K0 ? N0 : K0) > (N1 > K1 ? N1 : K1) ? (N0 > K0 ? N0 : K0) : (N1 > K1 ? N1 : K1);
    if (!gemm_tn_w128_serves(M, N0, K0, ldmax, ldmax, s)) return 0;
    return s;
}

extern "C" int vitk_gemm_tn_bf16_pair(const void* dY0, int64_t ldy0, const void* X0, int64_t ldx0, void* dW0, int64_t ldo0, int accumulate0,
                                      int64_t N0, int64_t K0, const void* dY1, int64_t ldy1, const void* X1, int64_t ldx1, void* dW1,
                                      int64_t ldo1, int accumulate1, int64_t N1, int64_t K1, int odt, int64_t M, float* ws, int64_t splits,
                                      void* stream) {
    if (!dY0 || !X0 || !dW0 || !dY1 || !X1 || !dW1 || !ws) VITK_FAIL(VITK_E_ARG, "gemm_tn_bf16_pair: null pointer");
    if (splits < 1 || splits != vitk_gemm_tn_pair_splits(M, N0, K0, N1, K1))
        VITK_FAIL(VITK_E_SHAPE, "gemm_tn_bf16_pair: splits must be vitk_gemm_tn_pair_splits(M, N0, K0, N1, K1) (0 = pair not served)");
    if ((ldy0 & 7) || (ldx0 & 7) || (ldo0 & 3) || (ldy1 & 7) || (ldx1 & 7) || (ldo1 & 3) || !aligned16(dY0) || !aligned16(X0) || !aligned16(dW0) ||
        !aligned16(dY1) || !aligned16(X1) || !aligned16(dW1) || !aligned16(ws))
        VITK_FAIL(VITK_E_ALIGN, "gemm_tn_bf16_pair: ldy/ldx %% 8, ldo %% 4 and 16-byte aligned pointers required");
    if (!gemm_tn_w128_serves(M, N0, K0, ldy0 > ldy1 ? ldy0 : ldy1, ldx0 > ldx1 ? ldx0 : ldx1, splits))
        VITK_FAIL(VITK_E_SHAPE, "gemm_tn_bf16_pair: row strides beyond the kernel's 32-bit offsets");
    const int rc = gemm_tn_w128_launch2(dY0, ldy0, X0, ldx0, N0, K0, dY1, ldy1, X1, ldx1, N1, K1, ws, M, splits, stream);
    if (rc != 0) return rc;
    hipStream_t st = (hipStream_t)stream;
    const long long NK0 = (long long)N0 * K0, NK1 = (long long)N1 * K1, stride = NK0 + NK1;
    VITK_DISPATCH_DT(odt, OT, hipLaunchKernelGGL((tn_reduce_kernel<OT>), dim3((unsigned)((NK0 / 4 + 255) / 256)), dim3(256), 0, st, ws, (int)splits, NK0,
                                                  (int)K0, (OT*)dW0, (long long)ldo0, accumulate0, stride));
    VITK_DISPATCH_DT(odt, OT, hipLaunchKernelGGL((tn_reduce_kernel<OT>), dim3((unsigned)((NK1 / 4 + 255) / 256)), dim3(256), 0, st, ws + NK0, (int)splits, NK1,
                                                  (int)K1, (OT*)dW1, (long long)ldo1, accumulate1, stride));
    VITK_CHECK_LAUNCH("gemm_tn_pair_reduce");
    return 0;
}

