// gemm_tn_dma.hip -- weight-gradient GEMM for gfx950:  dW[N,K] = dY[M,N]^T . X[M,K]  (autograd of nn.Linear: vit.py:20,23,44,47)
//
// The reduction index (token row m) is the slow index of BOTH operands, so fragments need transposed reads.  The first TN kernel
// (gemm_tn256_kernel, gemm_bf16.hip) staged tiles through registers into padded LDS rows; its ds_write pass (64 KiB per step at
// ~79 B/clk) sat exposed between the MFMA phases -- ~28 % of a step, 833 TF/s.  This kernel uses the NT kernels' pipeline:
//
//  * operands go HBM -> LDS by global_load_lds (no VGPR round trip, no ds_write) into FOUR stages of 32 token rows, three steps in
//    flight, counted vmcnt (never drained in the loop);
//  * ping-pong wave groups: waves 0-3 and 4-7 run the same stream one barrier slot apart (R = transpose-read fragments + issue DMA,
//    M = 16 MFMAs), so each SIMD's matrix pipe always has one wave feeding it;
//  * LDS image built for ds_read_b64_tr_b16: a DMA instruction fills 1 KiB = two 512-byte rows of one operand; pieces are laid
//    down 1088 bytes apart and the 32-byte windows of odd rows are swapped pairwise (an XOR on the per-lane SOURCE address and
//    on the read address): the 8 rows a 32-lane group reads then fall into 8 different bank windows (tools/lds_banks.py);
//  * token rows past the end of a split are read from a zero page, so they add nothing;
//  * a lane ends up with 4 consecutive k of one n (A operand = X fragment, B = dY fragment): 16-byte stores of the f32 partial tile
//    into its split slab; vitk_gemm_tn_bf16 then folds the slabs deterministically (tn_reduce_kernel, as before).
//
// Tile 256 (n) x 256 (k), 8 waves as 2 (n) x 4 (k), wave tile 128 x 64 = 8 x 4 fragments of 16 x 16.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int T_PIECE = 1088;                       // 1 KiB of data (2 rows x 512 B) + 64 B: consecutive pieces shift by 2 bank windows
constexpr int T_OPER_BYTES = 16 * T_PIECE;          // 32 rows of one operand
constexpr int T_STAGE_BYTES = 2 * T_OPER_BYTES;     // dY tile + X tile: 34,816
constexpr int T_LDS_BYTES = 4 * T_STAGE_BYTES;      // 139,264

#define TT_BARRIER() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

__device__ __attribute__((aligned(16))) char t_zero_page[512];      // zero-initialised: source of rows past the end of a split

__device__ __forceinline__ int t_xcd_swizzle(int b, int nwg) {
    const int q = nwg / 8, r = nwg % 8;
    const int xcd = b % 8, idx = b / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// byte offset of (row, byte-in-row) inside one operand's stage image
__device__ __forceinline__ int t_img(int row, int byte_in_row) {
    return (row >> 1) * T_PIECE + (row & 1) * 512 + (byte_in_row ^ ((row & 1) << 5));
}

// ds_read_b64_tr_b16 as inline asm: with the builtin, hipcc puts s_waitcnt vmcnt(0) in front of every group of transpose reads
// (it cannot tell them from the LDS-DMA writes in flight) and drains the DMA pipeline every K-step.  The asm form is invisible to
// that pass; the reads are retired by the explicit lgkmcnt(0) + sched_barrier that end every R slot.
__device__ __forceinline__ s16x4 t_tr(unsigned lds_addr) {
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(lds_addr) : "memory");
    return v;
}
__device__ __forceinline__ bf16x8 t_frag(unsigned stage, int off_lo, int off_hi) {
    // two transpose reads: token rows {4g..4g+3} and {16+4g..16+4g+3} of one 16-column block -> 8 k-values per lane
    const s16x4 lo = t_tr(stage + off_lo), hi = t_tr(stage + off_hi);
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// [measured] Moving the X tile off the LDS-DMA path (global_load -> registers -> ds_write, inline-asm loads counted by hand in the
// same vmcnt protocol) was built and removed: 0.84x (the "no MFMA" time rose from 237 to 307 us) -- the load side is not limited by
// the DMA path itself.  What bounds both TN kernels is the issue rate of ds_read_b64_tr_b16: a group's 16 transpose reads take ~405
// cycles to issue (tools/tn_stamps.py), against 272 for the 16 MFMAs they feed.
__global__ __launch_bounds__(512) void gemm_tn_dma_kernel(
    const __bf16* __restrict__ dY, long long ldy, const __bf16* __restrict__ X, long long ldx,
    float* __restrict__ ws, int M, int N, int K, int rows_per_split, int tiles_k, int nwg, int dbg, long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wk = wave & 3;       // wave tile: 128 (n) x 64 (k)
    const bool grp_b = wave >= 4;                  // = wn: the two n-halves are the two ping-pong groups
    // XCD-aware order over (split, tile) jointly, split-major: one XCD works on (mostly) ONE M-split, i.e. on workgroups that read
    // the same dY / X rows at the same time
    const int lin = t_xcd_swizzle(blockIdx.x, (int)gridDim.x);
    const int split = lin / nwg;
    const int wg = lin % nwg;
    const int tn = wg / tiles_k, tk = wg % tiles_k;
    const int n0 = tn * 256, k0 = tk * 256;
    const int mbeg = split * rows_per_split;
    int mend = mbeg + rows_per_split; mend = mend < M ? mend : M;
    const int nsteps = mend > mbeg ? (mend - mbeg + 31) / 32 : 0;

    // ---- producer: wave w fills pieces 2w, 2w+1 of each operand: piece p = token rows 2p, 2p+1 of the step.  Lane l sits at
    //      physical 16-byte chunk l & 31 of row l >> 5, which holds logical chunk (l & 31) ^ (2 * (row & 1)).
    const int prow = lane >> 5;
    const int pchunk = (lane & 31) ^ (prow << 1);
    const bool y_ok = n0 + pchunk * 8 < N, x_ok = k0 + pchunk * 8 < K;          // N, K are multiples of 8: a chunk is all or nothing
    const char* zsrc = t_zero_page + (lane & 31) * 16;
    auto issue = [&](int step, bool second) {
        char* base = lds + (step & 3) * T_STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int piece = wave * 2 + j;
            const int m = mbeg + step * 32 + piece * 2 + prow;
            const bool mv = m < mend;
            if (!second) {
                const char* src = (mv && y_ok) ? reinterpret_cast<const char*>(dY + (long long)m * ldy + n0 + pchunk * 8) : zsrc;
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                                 (void __attribute__((address_space(3)))*)(base + piece * T_PIECE), 16, 0, 0);
            } else {
                const char* src = (mv && x_ok) ? reinterpret_cast<const char*>(X + (long long)m * ldx + k0 + pchunk * 8) : zsrc;
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                                 (void __attribute__((address_space(3)))*)(base + T_OPER_BYTES + piece * T_PIECE), 16, 0, 0);
            }
        }
    };

    // ---- consumer: fragment addresses.  Lane (fi, fg): row 4 fg + (fi >> 2) (and + 16), bytes (fi & 3) * 8 of a 32-byte block.
    const int fi = lane & 15, fg = lane >> 4;
    const int r_lo = 4 * fg + (fi >> 2), r_hi = r_lo + 16;
    const int cb = (fi & 3) * 8;
    // dY: columns wn * 128 + f * 16 (f = 0..7) -> byte wn * 256 + f * 32 ; X: columns wk * 64 + f * 16 -> byte wk * 128 + f * 32
    int y_lo[8], y_hi[8], x_lo[4], x_hi[4];
#pragma unroll
    for (int f = 0; f < 8; ++f) { y_lo[f] = t_img(r_lo, wn * 256 + f * 32 + cb); y_hi[f] = t_img(r_hi, wn * 256 + f * 32 + cb); }
#pragma unroll
    for (int f = 0; f < 4; ++f) { x_lo[f] = T_OPER_BYTES + t_img(r_lo, wk * 128 + f * 32 + cb); x_hi[f] = T_OPER_BYTES + t_img(r_hi, wk * 128 + f * 32 + cb); }

    f32x4 acc[4][8];            // acc[fk][fn][j]: k = wk * 64 + 16 fk + 4 fg + j, n = wn * 128 + 16 fn + fi
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)lds);
    // experiments: s_memtime after every barrier of K-steps 8..39, waves 0 and 4 of workgroup 0 (VITK_TN_STAMPS)
    const bool stamp_on = stamps && blockIdx.x == 17 && lane == 0 && (wave == 0 || wave == 4);
    long long* my_stamps = stamps ? stamps + (wave == 4 ? 1024 : 0) : nullptr;
#define T_STAMP(t, slot) do { if (stamp_on && (t) >= 8 && (t) < 40) my_stamps[((t) - 8) * 8 + (slot)] = __builtin_readcyclecounter(); } while (0)
    auto body = [&](int t) __attribute__((always_inline)) {
        const unsigned base = lds_base + (t & 3) * T_STAGE_BYTES;
        const bool more = t + 3 < nsteps;
        bf16x8 xf[4], yf[4];
        T_STAMP(t, 0);
        // ---- R0: X fragments (kept for both halves) + dY fragments 0..3, DMA of the dY tile of step t + 3 ----
#pragma unroll
        for (int f = 0; f < 4; ++f) xf[f] = t_frag(base, x_lo[f], x_hi[f]);
#pragma unroll
        for (int f = 0; f < 4; ++f) yf[f] = t_frag(base, y_lo[f], y_hi[f]);
        T_STAMP(t, 1);
        if (more && !(dbg & 1)) issue(t + 3, false);
        T_STAMP(t, 2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        T_STAMP(t, 3);
        TT_BARRIER();
        T_STAMP(t, 4);
        // ---- M0 ----
        __builtin_amdgcn_s_setprio(1);
        if (dbg & 2) {
#pragma unroll
            for (int f = 0; f < 4; ++f) { asm volatile("" :: "v"(xf[f])); asm volatile("" :: "v"(yf[f])); }
        } else
#pragma unroll
        for (int fk = 0; fk < 4; ++fk)
#pragma unroll
            for (int f = 0; f < 4; ++f)
                acc[fk][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[fk], yf[f], acc[fk][f], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        T_STAMP(t, 5);
        TT_BARRIER();
        T_STAMP(t, 6);
        // ---- R1: dY fragments 4..7, X tile of step t + 3 (DMA or register loads), counted wait for step t + 1 ----
#pragma unroll
        for (int f = 0; f < 4; ++f) yf[f] = t_frag(base, y_lo[4 + f], y_hi[4 + f]);
        if (more && !(dbg & 1)) issue(t + 3, true);
        if (dbg & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (more) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (t + 2 < nsteps) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        T_STAMP(t, 7);
        TT_BARRIER();
        // ---- M1 ----
        __builtin_amdgcn_s_setprio(1);
        if (dbg & 2) {
#pragma unroll
            for (int f = 0; f < 4; ++f) asm volatile("" :: "v"(yf[f]));
        } else
#pragma unroll
        for (int fk = 0; fk < 4; ++fk)
#pragma unroll
            for (int f = 0; f < 4; ++f)
                acc[fk][4 + f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[fk], yf[f], acc[fk][4 + f], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        TT_BARRIER();
    };

    if (nsteps > 0) {
        // prologue: steps 0..2 in flight, step 0 landed
        issue(0, false); issue(0, true);
        if (nsteps > 1) { issue(1, false); issue(1, true); }
        if (nsteps > 2) { issue(2, false); issue(2, true); }
        if (nsteps > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (nsteps > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TT_BARRIER();
        if (grp_b) TT_BARRIER();   // group B runs one slot behind group A

        for (int t = 0; t < nsteps; ++t) body(t);
        if (!grp_b) TT_BARRIER();  // pairs with group B's extra barrier
    }

    // ---- partial tile -> ws[split][n][k]: 16 bytes per lane (4 consecutive k), 4 lanes = 64 contiguous bytes of a row n ----
    float* out = ws + (long long)split * N * K;
#pragma unroll
    for (int fn = 0; fn < 8; ++fn) {
        const int n = n0 + wn * 128 + fn * 16 + fi;
        if (n >= N) continue;
#pragma unroll
        for (int fk = 0; fk < 4; ++fk) {
            const int k = k0 + wk * 64 + fk * 16 + 4 * fg;
            if (k < K) *reinterpret_cast<f32x4*>(out + (long long)n * K + k) = acc[fk][fn];      // K % 4 == 0: 4 k's are all or nothing
        }
    }
}

}  // namespace

int gemm_tn_dma_launch(const void* dY, int64_t ldy, const void* X, int64_t ldx, float* ws, int64_t M, int64_t N, int64_t K,
                       int64_t splits, void* stream) {
    const int tiles_n = (int)((N + 255) / 256), tiles_k = (int)((K + 255) / 256);
    const int nwg = tiles_n * tiles_k;
    long long rps = (M + splits - 1) / splits;
    rps = (rps + 31) / 32 * 32;
    static const int rc__ = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_dma_kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS_BYTES);
    if (rc__ != 0) VITK_FAIL(rc__, "gemm_tn_bf16: cannot enable %d B of LDS", T_LDS_BYTES);
    hipLaunchKernelGGL(gemm_tn_dma_kernel, dim3((unsigned)(nwg * splits)), dim3(512), T_LDS_BYTES, (hipStream_t)stream,
                       (const __bf16*)dY, (long long)ldy, (const __bf16*)X, (long long)ldx, ws, (int)M, (int)N, (int)K, (int)rps, tiles_k, nwg,
                       getenv("VITK_TN_DBG") ? atoi(getenv("VITK_TN_DBG")) : 0,
                       getenv("VITK_TN_STAMPS") ? (long long*)strtoull(getenv("VITK_TN_STAMPS"), nullptr, 0) : (long long*)nullptr);
    VITK_CHECK_LAUNCH("gemm_tn_bf16 (dma)");
    return 0;
}
