// gemm_tn_w128.hip -- weight-gradient GEMM for gfx950, round 4:  dW[N,K] = dY[M,N]^T . X[M,K]  (autograd of nn.Linear: vit.py:20,23,44,47)
//
// 256 x 256 output tile per workgroup, split over the token rows M into f32 slabs (vitk_gemm_tn_bf16 folds them), like the two
// kernels before it (gemm_tn256_kernel in gemm_bf16.hip, gemm_tn_dma.hip) -- but FOUR waves, one per SIMD, each owning a
// 128 x 128 wave tile.  Why [measured, tools/tn_probe.hip, ViT-B/16 batch 256 shapes, same box]:
//   * the 8-wave kernels read 12 fragments per 32 MFMAs and were bound by the issue of ds_read_b64_tr_b16 (their transposing reads
//     and their MFMAs did not overlap: reads alone 150 us + MFMA alone ~113 us ~ the complete 273 us); a 128 x 128 wave tile reads
//     16 fragments per 64 MFMAs;
//   * with one wave per SIMD the register file has room for the 64 accumulator tiles (256 AGPRs) AND two full fragment sets (2 x 16
//     fragments = 128 VGPRs): the fragments of step t + 1 are read during the MFMAs of step t, one instruction per MFMA gap, and no
//     read is waited for right after its issue -- no ping-pong partner needed;
//   * the accumulators are PINNED: the MFMAs are asm statements with "+a" operands.  With the builtin and 128 fragment registers live
//     hipcc moved accumulator tuples between the AGPR and VGPR files around every MFMA (148 v_accvgpr_write + 68 _read + 65 s_nop
//     per two steps; with a branch between the two halves of the unrolled pair it kept the fragment sets in scratch).  asm volatile
//     statements keep their program order, so the step is issued exactly as written: per step 64 MFMAs, 32 transposing reads, 8
//     LDS-DMA instructions, one barrier;
//   * operands by LDS-DMA through BUFFER DESCRIPTORS (buffer_load_dwordx4 ... lds): a per-lane 32-bit offset formed once + a scalar
//     step offset, i.e. no per-piece address arithmetic in the loop (the global_load_lds form cost ~10 VALU per piece), and token
//     rows past the end of the split are outside the descriptor's range and arrive as zeros (no zero page, no predicates).
// Result: dW1 (3072 x 768, M = 50,432) 304 -> 200 us with the fold (783 -> 1,190 TF/s), kernel alone 180 us = 1,322 TF/s; ablations
// of the same loop: MFMA only 148, DMA only 157, reads only 51, empty loop 17 us -- what is left is the LDS-DMA feed (three steps =
// 96 KiB in flight per CU against ~2 us of loaded HBM latency) beside the MFMA stream.  Results are bit-identical to the 8-wave
// kernels' (same products, same order of accumulation per output element).
//
// LDS image, pieces and swizzle are gemm_tn_dma.hip's: 4 stages of 32 token rows; a DMA instruction fills 1 KiB = two 512-byte rows
// of one operand; pieces lie 1088 bytes apart and the 32-byte windows of odd rows are swapped pairwise (XOR on the per-lane source
// chunk and on the read address), so the 8 rows a 32-lane group reads fall into 8 different bank windows.
#include "common.h"
#include <stdlib.h>

#ifdef VITK_HALF_IS_F16
#define VITK_MFMA_ASM "v_mfma_f32_16x16x32_f16"
#else
#define VITK_MFMA_ASM "v_mfma_f32_16x16x32_bf16"
#endif

namespace {

constexpr int W_PIECE = 1088;                       // 1 KiB of data (2 rows x 512 B) + 64 B
constexpr int W_OPER_BYTES = 16 * W_PIECE;          // 32 rows of one operand
constexpr int W_STAGE_BYTES = 2 * W_OPER_BYTES;     // 34,816
constexpr int W_LDS_BYTES = 4 * W_STAGE_BYTES;      // 139,264

__device__ __forceinline__ int w_xcd_swizzle(int b, int nwg) {
    const int q = nwg / 8, r = nwg % 8;
    const int xcd = b % 8, idx = b / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
#define W_PIN() __builtin_amdgcn_sched_barrier(0)

template <int OFF> __device__ __forceinline__ s16x4 w_tr(unsigned lds_addr) {
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(OFF) : "memory");
    return v;
}
// fragment f (16 columns = 32 bytes) of a wave's 128 columns: two transposing reads (token rows 4g..4g+3 and 16+4g..16+4g+3).
// Odd token rows have their 32-byte windows swapped pairwise (the image's bank swizzle): f -> f ^ 1 there, which is folded into
// two per-lane bases (even f / odd f) so that the fragment index itself is an immediate.
template <int F> __device__ __forceinline__ bf16x8 w_frag(unsigned b_e, unsigned b_o) {
    const s16x4 lo = w_tr<F * 32>((F & 1) ? b_o : b_e), hi = w_tr<F * 32 + 8 * W_PIECE>((F & 1) ? b_o : b_e);
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// MFMA as asm with the accumulator PINNED to AGPRs ("+a"): with the builtin and 128 fragment registers live hipcc moved accumulator
// tuples between the two files around every MFMA (148 v_accvgpr_write + 68 _read + 65 s_nop per two steps).  asm volatile statements
// keep their program order, so the K-step below is issued exactly as written.
__device__ __forceinline__ void w_mfma(f32x4& c, const bf16x8& a, const bf16x8& b) {
    asm volatile("" VITK_MFMA_ASM " %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ bf16x8 w_join(s16x4 lo, s16x4 hi) {
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// One or TWO problems per launch (vitk_gemm_tn_bf16_pair): the tiles of both share the split count and the token rows M, so that e.g. the
// weight gradients of to_qkv (27 tiles at ViT-B/16) and of to_out (9 tiles) fill the chip as 36 tiles x 7 splits -- 252 f32 slabs instead
// of 27 x 9 + 9 x 28 = 495, one launch + one fold less per layer.  A split's slab holds problem 0's N0 x K0 floats, then problem 1's.
struct TnProblem { const __bf16* dY; long long ldy; const __bf16* X; long long ldx; int N, K, tiles_k, nwg; };

template <int ABL>
__global__ __launch_bounds__(256) void gemm_tn_w128_kernel(const TnProblem p0, const TnProblem p1, float* __restrict__ ws, int M, int rows_per_split,
                                                           long long slab_stride) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wk = wave & 1;        // wave tile: 128 (n) x 128 (k)
    const int lin = w_xcd_swizzle(blockIdx.x, (int)gridDim.x);
    const int nwg_all = p0.nwg + p1.nwg;
    const int split = lin / nwg_all;
    int wg = lin % nwg_all;
    const bool second = wg >= p0.nwg;               // wave-uniform: every field below lives in SGPRs
    if (second) wg -= p0.nwg;
    const __bf16* __restrict__ dY = second ? p1.dY : p0.dY;
    const __bf16* __restrict__ X = second ? p1.X : p0.X;
    const long long ldy = second ? p1.ldy : p0.ldy, ldx = second ? p1.ldx : p0.ldx;
    const int N = second ? p1.N : p0.N, K = second ? p1.K : p0.K, tiles_k = second ? p1.tiles_k : p0.tiles_k;
    const int tn = wg / tiles_k, tk = wg % tiles_k;
    const int n0 = tn * 256, k0 = tk * 256;
    const int mbeg = split * rows_per_split;
    int mend = mbeg + rows_per_split; mend = mend < M ? mend : M;
    const int R = mend > mbeg ? mend - mbeg : 0;
    const int nsteps = (R + 31) / 32;

    // ---- producer: wave w fills pieces 4w .. 4w + 3 of each operand (piece p = token rows 2p, 2p + 1 of the step) ----
    const int prow = lane >> 5;
    const int pchunk = (lane & 31) ^ (prow << 1);
    // descriptors: base = element (mbeg, n0) / (mbeg, k0); the range ends with the last valid element of the split's last row, so
    // rows >= R (and the columns past N of the last row) read as zeros; columns past N of other rows read the next row's first
    // elements: they only reach output columns >= N, which are never stored
    const long long ybytes = R > 0 ? ((long long)(R - 1) * ldy + (N - n0)) * 2 : 0;
    const long long xbytes = R > 0 ? ((long long)(R - 1) * ldx + (K - k0)) * 2 : 0;
    const auto yrs = __builtin_amdgcn_make_buffer_rsrc((void*)(dY + (long long)mbeg * ldy + n0), 0, (int)ybytes, 0x00020000);
    const auto xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(X + (long long)mbeg * ldx + k0), 0, (int)xbytes, 0x00020000);
    int yvo[4], xvo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (wave * 4 + j) * 2 + prow;
        yvo[j] = (int)(((long long)r * ldy + pchunk * 8) * 2);
        xvo[j] = (int)(((long long)r * ldx + pchunk * 8) * 2);
    }
    const int ystep = (int)(64 * ldy), xstep = (int)(64 * ldx);        // bytes per 32 token rows
    // piece q (0..7) of step `step`: q < 4 dY piece 4w + q, else X piece 4w + q - 4
    auto dma = [&](int step, int q) __attribute__((always_inline)) {
        char* dst = lds + (step & 3) * W_STAGE_BYTES + (q >= 4 ? W_OPER_BYTES : 0) + (wave * 4 + (q & 3)) * W_PIECE;
        if (q < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(yrs, (void __attribute__((address_space(3)))*)dst, 16, yvo[q & 3], step * ystep, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (void __attribute__((address_space(3)))*)dst, 16, xvo[q & 3], step * xstep, 0, 0);
    };

    // ---- consumer: lane (fi, fg) reads token row 4 fg + (fi >> 2) (and + 16), bytes (fi & 3) * 8 of a 32-byte window ----
    const int fi = lane & 15, fg = lane >> 4;
    const int r_lo = 4 * fg + (fi >> 2);          // r_hi = r_lo + 16: same parity, 8 pieces further
    const int odd = r_lo & 1;
    const int rowb = (r_lo >> 1) * W_PIECE + odd * 512 + (fi & 3) * 8;
    // per-lane bases without the stage: [operand][even f / odd f]; hi = lo + 8 * W_PIECE
    const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)lds);
    const unsigned yb_e = lds_base + rowb + wn * 256 + odd * 32, yb_o = lds_base + rowb + wn * 256 - odd * 32;
    const unsigned xb_e = yb_e - wn * 256 + wk * 256 + W_OPER_BYTES, xb_o = yb_o - wn * 256 + wk * 256 + W_OPER_BYTES;

    f32x4 acc[8][8];            // acc[fk][fn][j]: k = wk * 128 + 16 fk + 4 fg + j, n = wn * 128 + 16 fn + fi
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // all 16 fragments of one stage into (XF, YF)
#define W_READ_ONE(G, XF, YF, SOFF) do { \
        if constexpr ((G) < 8) XF[(G)] = w_frag<(G)>(xb_e + (SOFF), xb_o + (SOFF)); \
        else YF[(G) - 8] = w_frag<(G) - 8>(yb_e + (SOFF), yb_o + (SOFF)); \
    } while (0)

    // one step: 16 groups of {4 MFMAs on the current fragments, one fragment (two transposing reads) of the next step, every second
    // group one DMA piece of step t + 4 (into the stage this step's fragments were read from: free since the barrier that ended
    // step t - 1)}, one instruction per MFMA gap
#define W_GROUP(G, T_, XC, YC, XN, YN, SOFF) do { \
        constexpr int fk_ = (G) >> 1, h_ = (G) & 1; \
        constexpr int F_ = (G) & 7; \
        const unsigned ra_ = ((G) < 8 ? ((F_ & 1) ? xb_o : xb_e) : ((F_ & 1) ? yb_o : yb_e)) + (SOFF); \
        s16x4 lo_ = {0, 0, 0, 0}, hi_ = {0, 0, 0, 0}; \
        if constexpr (!(ABL & 4)) w_mfma(acc[fk_][h_ * 4 + 0], XC[fk_], YC[h_ * 4 + 0]); \
        if constexpr (!(ABL & 2)) lo_ = w_tr<F_ * 32>(ra_); \
        if constexpr (!(ABL & 4)) w_mfma(acc[fk_][h_ * 4 + 1], XC[fk_], YC[h_ * 4 + 1]); \
        if constexpr (!(ABL & 2)) hi_ = w_tr<F_ * 32 + 8 * W_PIECE>(ra_); \
        if constexpr (!(ABL & 4)) w_mfma(acc[fk_][h_ * 4 + 2], XC[fk_], YC[h_ * 4 + 2]); \
        if constexpr (!(ABL & 1) && ((G) & 1)) { W_PIN(); dma((T_) + 4, (G) >> 1); W_PIN(); } \
        if constexpr (!(ABL & 4)) w_mfma(acc[fk_][h_ * 4 + 3], XC[fk_], YC[h_ * 4 + 3]); \
        if constexpr (!(ABL & 2)) { if constexpr ((G) < 8) XN[F_] = w_join(lo_, hi_); else YN[F_] = w_join(lo_, hi_); } \
    } while (0)
#define W_STEP(T_, XC, YC, XN, YN) do { \
        const int t_ = (T_); \
        const unsigned soff = ((t_ + 1) & 3) * W_STAGE_BYTES; \
        __builtin_amdgcn_s_setprio(1); \
        W_GROUP(0, t_, XC, YC, XN, YN, soff); W_GROUP(1, t_, XC, YC, XN, YN, soff); W_GROUP(2, t_, XC, YC, XN, YN, soff); W_GROUP(3, t_, XC, YC, XN, YN, soff); \
        W_GROUP(4, t_, XC, YC, XN, YN, soff); W_GROUP(5, t_, XC, YC, XN, YN, soff); W_GROUP(6, t_, XC, YC, XN, YN, soff); W_GROUP(7, t_, XC, YC, XN, YN, soff); \
        W_GROUP(8, t_, XC, YC, XN, YN, soff); W_GROUP(9, t_, XC, YC, XN, YN, soff); W_GROUP(10, t_, XC, YC, XN, YN, soff); W_GROUP(11, t_, XC, YC, XN, YN, soff); \
        W_GROUP(12, t_, XC, YC, XN, YN, soff); W_GROUP(13, t_, XC, YC, XN, YN, soff); W_GROUP(14, t_, XC, YC, XN, YN, soff); W_GROUP(15, t_, XC, YC, XN, YN, soff); \
        __builtin_amdgcn_s_setprio(0); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      /* the next step's fragments are in registers */ \
        if constexpr (!(ABL & 1)) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   /* own pieces of step t + 2 landed (t + 3, t + 4 fly) */ \
        W_PIN(); \
        __builtin_amdgcn_s_barrier();           /* stage t + 2 visible to all, stage t + 1 read by all */ \
        W_PIN(); \
    } while (0)

    if (nsteps > 0) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int q = 0; q < 8; ++q) dma(s, q);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");       // steps 0, 1 landed
        W_PIN();
        __builtin_amdgcn_s_barrier();
        W_PIN();
        bf16x8 xa[8], ya[8], xb[8], yb[8];
        W_READ_ONE(0, xa, ya, 0u); W_READ_ONE(1, xa, ya, 0u); W_READ_ONE(2, xa, ya, 0u); W_READ_ONE(3, xa, ya, 0u);
        W_READ_ONE(4, xa, ya, 0u); W_READ_ONE(5, xa, ya, 0u); W_READ_ONE(6, xa, ya, 0u); W_READ_ONE(7, xa, ya, 0u);
        W_READ_ONE(8, xa, ya, 0u); W_READ_ONE(9, xa, ya, 0u); W_READ_ONE(10, xa, ya, 0u); W_READ_ONE(11, xa, ya, 0u);
        W_READ_ONE(12, xa, ya, 0u); W_READ_ONE(13, xa, ya, 0u); W_READ_ONE(14, xa, ya, 0u); W_READ_ONE(15, xa, ya, 0u);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        W_PIN();
        __builtin_amdgcn_s_barrier();           // stage 0 has been read by everyone: step 0 may refill it
        W_PIN();
        if constexpr (ABL & 2) {
#pragma unroll
            for (int f = 0; f < 8; ++f) { xb[f] = xa[f]; yb[f] = ya[f]; }
        }
        // steps in PAIRS, unconditionally (the two fragment sets swap roles; a branch between the halves made hipcc keep them in
        // scratch): an odd count runs one more step on a stage the out-of-range DMA filled with zeros
        for (int t = 0; t < nsteps; t += 2) {
            W_STEP(t, xa, ya, xb, yb);
            W_STEP(t + 1, xb, yb, xa, ya);
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // nothing may land in this LDS allocation after the workgroup is gone; the asm MFMAs' results are complete before the compiler's reads of them
    }
#undef W_STEP
#undef W_GROUP
#undef W_READ_ONE

    // partial tile -> ws[split][n][k]: 16 bytes per lane (4 consecutive k)
    float* out = ws + (long long)split * slab_stride + (second ? (long long)p0.N * p0.K : 0);
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

}  // namespace

int gemm_tn_w128_launch2(const void* dY0, int64_t ldy0, const void* X0, int64_t ldx0, int64_t N0, int64_t K0, const void* dY1, int64_t ldy1,
                         const void* X1, int64_t ldx1, int64_t N1, int64_t K1, float* ws, int64_t M, int64_t splits, void* stream);

// The descriptor offsets are 32-bit: (rows of a split + the 4 steps the DMA stream runs past its end) x row stride must stay below 2^31.
bool gemm_tn_w128_serves(int64_t M, int64_t N, int64_t K, int64_t ldy, int64_t ldx, int64_t splits) {
    long long rps = (M + splits - 1) / splits;
    rps = (rps + 31) / 32 * 32;
    const long long ldmax = ldy > ldx ? ldy : ldx;
    return (rps + 160) * ldmax * 2 < (1LL << 31);
}

int gemm_tn_w128_launch(const void* dY, int64_t ldy, const void* X, int64_t ldx, float* ws, int64_t M, int64_t N, int64_t K,
                        int64_t splits, void* stream) {
    return gemm_tn_w128_launch2(dY, ldy, X, ldx, N, K, nullptr, 0, nullptr, 0, 0, 0, ws, M, splits, stream);
}

// two problems (the second may be absent: dY1 == nullptr) behind one split count; slab of a split = [N0 * K0 | N1 * K1] floats
int gemm_tn_w128_launch2(const void* dY0, int64_t ldy0, const void* X0, int64_t ldx0, int64_t N0, int64_t K0, const void* dY1, int64_t ldy1,
                         const void* X1, int64_t ldx1, int64_t N1, int64_t K1, float* ws, int64_t M, int64_t splits, void* stream) {
    auto mk = [](const void* dY, int64_t ldy, const void* X, int64_t ldx, int64_t N, int64_t K) {
        TnProblem p{};
        p.dY = (const __bf16*)dY; p.ldy = ldy; p.X = (const __bf16*)X; p.ldx = ldx; p.N = (int)N; p.K = (int)K;
        p.tiles_k = dY ? (int)((K + 255) / 256) : 1;
        p.nwg = dY ? (int)((N + 255) / 256) * p.tiles_k : 0;
        return p;
    };
    const TnProblem p0 = mk(dY0, ldy0, X0, ldx0, N0, K0), p1 = mk(dY1, ldy1, X1, ldx1, N1, K1);
    long long rps = (M + splits - 1) / splits;
    rps = (rps + 31) / 32 * 32;
    static const int rc__ = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_w128_kernel<0>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS_BYTES);
    if (rc__ != 0) VITK_FAIL(rc__, "gemm_tn_bf16: cannot enable %d B of LDS", W_LDS_BYTES);
    hipLaunchKernelGGL(gemm_tn_w128_kernel<0>, dim3((unsigned)((p0.nwg + p1.nwg) * splits)), dim3(256), W_LDS_BYTES, (hipStream_t)stream, p0, p1, ws,
                       (int)M, (int)rps, (long long)(N0 * K0 + (dY1 ? N1 * K1 : 0)));
    VITK_CHECK_LAUNCH("gemm_tn_bf16 (w128)");
    return 0;
}
