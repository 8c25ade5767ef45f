// tools/tn_probe.hip -- round-4 experiment bench for the weight-gradient GEMM dW[N,K] = dY[M,N]^T X[M,K] (not part of libvitk).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/tn_probe.hip -o tools/tn_probe.bin -ldl
//   tools/tn_probe.bin [M N K]      (run from the repo root; A/B against vit_pytorch_amd/libvitk.so's vitk_gemm_tn_bf16)
//
// Kernel under test: 256 x 256 output tile, FOUR waves (one per SIMD) with 128 x 128 wave tiles -- 64 accumulator tiles = 256 AGPRs,
// fragments DOUBLE-BUFFERED in VGPRs (2 x 16 fragments = 128 registers: with one wave per SIMD the file has room), so the
// transposing reads of step t + 1 are spread evenly under the 64 MFMAs of step t (one fragment after every four MFMAs) and no
// read is ever waited for right after its issue; operands by LDS-DMA through buffer descriptors (buffer_load ... lds: a per-lane
// 32-bit offset formed once + a scalar step offset; rows past the end of the split are out of the descriptor's range and arrive
// as zeros -- no zero page, no per-piece address arithmetic), one DMA piece after every eighth MFMA; one barrier per step.
// Ablations (timing only, results are wrong): bit 0 no DMA in the loop, bit 1 no transposing reads, bit 2 no MFMA.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int T_PIECE_PAD = 1088;                   // PAD = 1: 1 KiB of data (2 rows x 512 B) + 64 B
template <int PAD> struct Geo {
    static constexpr int PIECE = PAD ? T_PIECE_PAD : 1024;
    static constexpr int OPER = 16 * PIECE;          // 32 rows of one operand
    static constexpr int STAGE = 2 * OPER;           // 34,816 / 32,768
};

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
// MFMA as asm with the accumulator PINNED to AGPRs ("+a"): with the builtin and 128 fragment registers live hipcc moved accumulator
// tuples between the two files around every MFMA (148 v_accvgpr_write + 68 _read + 65 s_nop per two steps).  asm volatile statements
// keep their program order, so the K-step below is issued exactly as written.
__device__ __forceinline__ void w_mfma(f32x4& c, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ bf16x8 w_join(s16x4 lo, s16x4 hi) {
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
template <int N_> __device__ __forceinline__ void w_wait_vm() {
    if constexpr (N_ == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N_ == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N_ == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if constexpr (N_ == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if constexpr (N_ == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else static_assert(N_ < 0, "unsupported vmcnt");
}

// NST stages of 32 token rows (NST - 1 steps in flight while one is read).
// PAD = 1: gemm_tn_dma.hip's image (pieces 1088 B apart, 32-byte windows of odd rows swapped pairwise).
// PAD = 0: pieces 1 KiB apart (5 stages = exactly the 160 KiB of a CU); the 32-byte window w of token row r sits at window
//          w ^ (r & 7) of its 256-byte half row, so the 8 rows a 32-lane group reads fall into 8 different bank windows.
// ST (store-coupling experiment for the NT epilogue question: do trickled global stores cost the LDS-DMA stream anything when the
// waves that store never wait on vmcnt?): 0 = none; 1 = coupled: every wave issues its 8 DMA pieces AND 2 stores (1 KiB each) per
// step, its counted wait allows for the younger stores; 2 = decoupled: waves 0, 1 issue all 32 DMA pieces of a step (16 each),
// waves 2, 3 issue 4 stores each and never wait on vmcnt.  8 KiB of stores per step and CU either way (the rate at which an NT tile's
// 128 KiB of 16-bit output would trickle out over 16 K-steps).
template <int ABL, int NST, int PAD, int ST = 0, int DPL = 0>
__global__ __launch_bounds__(256) void tn_w128_kernel(
    const __bf16* __restrict__ dY, long long ldy, const __bf16* __restrict__ X, long long ldx,
    float* __restrict__ ws, int M, int N, int K, int rows_per_split, int tiles_k, int nwg, char* __restrict__ scratch, long long scratch_per_wave) {
    using G_ = Geo<PAD>;
    constexpr int PIECE = G_::PIECE, OPER = G_::OPER, STAGE = G_::STAGE;
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
    const int R = mend > mbeg ? mend - mbeg : 0;
    const int nsteps = (R + 31) / 32;

    // ---- producer: wave w fills pieces 4w .. 4w + 3 of each operand (piece p = token rows 2p, 2p + 1 of the step) ----
    // descriptors: base = element (mbeg, n0) / (mbeg, k0); the range ends with the last valid element of the split's last row, so
    // rows >= R (and the columns past N of the last row) read as zeros; columns past N of other rows read the next row's first
    // elements: they only reach output columns >= N, which are never stored
    const int prow = lane >> 5;
    const long long ybytes = R > 0 ? ((long long)(R - 1) * ldy + (N - n0)) * 2 : 0;
    const long long xbytes = R > 0 ? ((long long)(R - 1) * ldx + (K - k0)) * 2 : 0;
    const auto yrs = __builtin_amdgcn_make_buffer_rsrc((void*)(dY + (long long)mbeg * ldy + n0), 0, (int)ybytes, 0x00020000);
    const auto xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(X + (long long)mbeg * ldx + k0), 0, (int)xbytes, 0x00020000);
    int yvo[4], xvo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (wave * 4 + j) * 2 + prow;          // token row of the step; r & 7 = 2 j + prow
        const int pchunk = PAD ? ((lane & 31) ^ (prow << 1)) : ((lane & 31) ^ (((2 * j + prow) & 7) << 1));
        yvo[j] = (int)(((long long)r * ldy + pchunk * 8) * 2);
        xvo[j] = (int)(((long long)r * ldx + pchunk * 8) * 2);
    }
    const int ystep = (int)(64 * ldy), xstep = (int)(64 * ldx);        // bytes per 32 token rows
    // piece q (0..7) of step `step` into stage `stg`: q < 4 dY piece 4w + q, else X piece 4w + q - 4
    // dw = 0: this wave's own pieces; dw = 2 (ST = 2, waves 0 and 1): the pieces of wave + 2 (8 token rows further)
    auto dma = [&](int step, int stg, int q, int dw) __attribute__((always_inline)) {
        char* dst = lds + stg * STAGE + (q >= 4 ? OPER : 0) + ((wave + dw) * 4 + (q & 3)) * PIECE;
        if (q < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(yrs, (void __attribute__((address_space(3)))*)dst, 16, yvo[q & 3], step * ystep + dw * 8 * (int)ldy * 2, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (void __attribute__((address_space(3)))*)dst, 16, xvo[q & 3], step * xstep + dw * 8 * (int)ldx * 2, 0, 0);
    };
    char* sp = scratch + ((long long)blockIdx.x * 4 + wave) * scratch_per_wave + lane * 16;
    auto store1k = [&](const bf16x8& v) __attribute__((always_inline)) {
        asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(sp), "v"(v) : "memory");
        sp += 1024;
    };

    // ---- consumer: lane (fi, fg) reads token row 4 fg + (fi >> 2) (and + 16), bytes (fi & 3) * 8 of a 32-byte window ----
    const int fi = lane & 15, fg = lane >> 4;
    const int r_lo = 4 * fg + (fi >> 2);          // r_hi = r_lo + 16: same r & 7, 8 pieces further
    const int odd = r_lo & 1;
    const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)lds);
    const int rowb = (r_lo >> 1) * PIECE + odd * 512 + (fi & 3) * 8;
    // PAD = 1: two per-lane bases per operand (even f / odd f), the fragment index is an immediate
    const unsigned yb_e = lds_base + rowb + wn * 256 + odd * 32, yb_o = lds_base + rowb + wn * 256 - odd * 32;
    const unsigned xb_e = yb_e - wn * 256 + wk * 256 + OPER, xb_o = yb_o - wn * 256 + wk * 256 + OPER;
    // PAD = 0: per-lane window offsets fo[f] = (f ^ (r & 7)) * 32, one base per operand
    unsigned fo[8];
#pragma unroll
    for (int f = 0; f < 8; ++f) fo[f] = (unsigned)((f ^ (r_lo & 7)) * 32);
    const unsigned yb0 = lds_base + rowb + wn * 256, xb0 = lds_base + rowb + wk * 256 + OPER;

    f32x4 acc[8][8];            // acc[fk][fn][j]: k = wk * 128 + 16 fk + 4 fg + j, n = wn * 128 + 16 fn + fi
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // address of fragment G (0..7: X fragment G, 8..15: dY fragment G - 8) in the stage at byte offset SOFF; the two transposing reads
    // sit at immediates W_IMM_LO / W_IMM_HI of it
#define W_ADDR(G, SOFF) (PAD ? ((((G) < 8) ? ((((G) & 7) & 1) ? xb_o : xb_e) : ((((G) & 7) & 1) ? yb_o : yb_e)) + (SOFF)) \
                             : ((((G) < 8) ? xb0 : yb0) + (SOFF) + fo[(G) & 7]))
#define W_IMM_LO(G) (PAD ? ((G) & 7) * 32 : 0)
#define W_IMM_HI(G) (PAD ? ((G) & 7) * 32 + 8 * T_PIECE_PAD : 8 * 1024)
#define W_READ_ONE(G, XF, YF, SOFF) do { \
        const unsigned ra_ = W_ADDR(G, SOFF); \
        const s16x4 lo_ = w_tr<W_IMM_LO(G)>(ra_), hi_ = w_tr<W_IMM_HI(G)>(ra_); \
        if constexpr ((G) < 8) XF[(G) & 7] = w_join(lo_, hi_); else YF[(G) & 7] = w_join(lo_, hi_); \
    } while (0)

    // one step: 16 groups of {4 MFMAs on the current fragments, one fragment (two transposing reads) of the next step, every second
    // group one DMA piece of step t + NST (into the stage this step's fragments were read from: free since the barrier that ended
    // step t - 1)}, one instruction per MFMA gap
#define W_GROUP(G, T_, SCUR, XC, YC, XN, YN, SOFF) do { \
        constexpr int fk_ = (G) >> 1, h_ = (G) & 1; \
        constexpr int F_ = (G) & 7; \
        const unsigned ra_ = W_ADDR(G, SOFF); \
        s16x4 lo_ = {0, 0, 0, 0}, hi_ = {0, 0, 0, 0}; \
        if constexpr (!(ABL & 4)) w_mfma(acc[fk_][h_ * 4 + 0], XC[fk_], YC[h_ * 4 + 0]); \
        if constexpr (!(ABL & 2)) lo_ = w_tr<W_IMM_LO(G)>(ra_); \
        if constexpr (!(ABL & 4)) w_mfma(acc[fk_][h_ * 4 + 1], XC[fk_], YC[h_ * 4 + 1]); \
        if constexpr (!(ABL & 2)) hi_ = w_tr<W_IMM_HI(G)>(ra_); \
        if constexpr (!(ABL & 4)) w_mfma(acc[fk_][h_ * 4 + 2], XC[fk_], YC[h_ * 4 + 2]); \
        if constexpr (ROLE == 0 && ST == 1 && ((G) == 0 || (G) == 8)) store1k(XC[0]); \
        if constexpr (ROLE == 2 && ((G) & 3) == 0) store1k(XC[0]); \
        /* DPL: where the 8 DMA pieces of a step sit -- 0: after every second group; 1: groups 0-7; 2: groups 8-15; 3: as 0, no s_setprio */ \
        if constexpr (!(ABL & 1) && ROLE == 0 && (DPL == 0 || DPL == 3) && ((G) & 1)) { W_PIN(); dma((T_) + NST, SCUR, (G) >> 1, 0); W_PIN(); } \
        if constexpr (!(ABL & 1) && ROLE == 0 && DPL == 1 && (G) < 8) { W_PIN(); dma((T_) + NST, SCUR, (G), 0); W_PIN(); } \
        if constexpr (!(ABL & 1) && ROLE == 0 && DPL == 2 && (G) >= 8) { W_PIN(); dma((T_) + NST, SCUR, (G) - 8, 0); W_PIN(); } \
        if constexpr (!(ABL & 1) && ROLE == 1) { W_PIN(); dma((T_) + NST, SCUR, (G) & 7, 2 * ((G) >> 3)); W_PIN(); } \
        if constexpr (!(ABL & 4)) w_mfma(acc[fk_][h_ * 4 + 3], XC[fk_], YC[h_ * 4 + 3]); \
        if constexpr (!(ABL & 2)) { if constexpr ((G) < 8) XN[F_] = w_join(lo_, hi_); else YN[F_] = w_join(lo_, hi_); } \
    } while (0)
    // scur = stage of step t (its fragments are in registers; it is being refilled), snext = stage of step t + 1 (being read)
#define W_STEP(T_, XC, YC, XN, YN) do { \
        const int t_ = (T_); \
        const int snext_ = scur + 1 == NST ? 0 : scur + 1; \
        const unsigned soff = snext_ * STAGE; \
        if constexpr (DPL != 3) __builtin_amdgcn_s_setprio(1); \
        W_GROUP(0, t_, scur, XC, YC, XN, YN, soff); W_GROUP(1, t_, scur, XC, YC, XN, YN, soff); W_GROUP(2, t_, scur, XC, YC, XN, YN, soff); W_GROUP(3, t_, scur, XC, YC, XN, YN, soff); \
        W_GROUP(4, t_, scur, XC, YC, XN, YN, soff); W_GROUP(5, t_, scur, XC, YC, XN, YN, soff); W_GROUP(6, t_, scur, XC, YC, XN, YN, soff); W_GROUP(7, t_, scur, XC, YC, XN, YN, soff); \
        W_GROUP(8, t_, scur, XC, YC, XN, YN, soff); W_GROUP(9, t_, scur, XC, YC, XN, YN, soff); W_GROUP(10, t_, scur, XC, YC, XN, YN, soff); W_GROUP(11, t_, scur, XC, YC, XN, YN, soff); \
        W_GROUP(12, t_, scur, XC, YC, XN, YN, soff); W_GROUP(13, t_, scur, XC, YC, XN, YN, soff); W_GROUP(14, t_, scur, XC, YC, XN, YN, soff); W_GROUP(15, t_, scur, XC, YC, XN, YN, soff); \
        if constexpr (DPL != 3) __builtin_amdgcn_s_setprio(0); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      /* the next step's fragments are in registers */ \
        if constexpr (!(ABL & 1) && ROLE == 0) w_wait_vm<(ST == 1 ? 10 : 8) * (NST - 2)>();   /* own pieces of step t + 2 landed (t + 3 .. t + NST fly) */ \
        if constexpr (!(ABL & 1) && ROLE == 1) w_wait_vm<16 * (NST - 2)>(); \
        W_PIN(); \
        __builtin_amdgcn_s_barrier();           /* stage of t + 2 visible to all, stage of t + 1 read by all */ \
        W_PIN(); \
        scur = snext_; \
    } while (0)

    auto run = [&](auto role_c) __attribute__((always_inline)) {
        constexpr int ROLE = decltype(role_c)::value;       // 0: DMA + compute (+ stores when ST = 1); 1: all the DMA; 2: stores only
        if constexpr (ROLE != 2) {
#pragma unroll
            for (int s = 0; s < NST; ++s)
#pragma unroll
                for (int q = 0; q < (ROLE == 1 ? 16 : 8); ++q) dma(s, s, q & 7, 2 * (q >> 3));
            w_wait_vm<(ROLE == 1 ? 16 : 8) * (NST - 2)>();             // steps 0, 1 landed
        }
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
        int scur = 0;
        for (int t = 0; t < nsteps; t += 2) {
            W_STEP(t, xa, ya, xb, yb);
            W_STEP(t + 1, xb, yb, xa, ya);
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // nothing may land in this LDS allocation after the workgroup is gone; the asm MFMAs' results are complete before the compiler's reads of them
    };
    if (nsteps > 0) {
        if constexpr (ST == 2) {
            if (wave < 2) run(std::integral_constant<int, 1>{}); else run(std::integral_constant<int, 2>{});
        } else run(std::integral_constant<int, 0>{});
    }
#undef W_STEP
#undef W_GROUP
#undef W_READ_ONE

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

typedef void (*kern_t)(const __bf16*, long long, const __bf16*, long long, float*, int, int, int, int, int, int, char*, long long);

int main(int argc, char** argv) {
    const long long M = argc > 3 ? atoll(argv[1]) : 50432, N = argc > 3 ? atoll(argv[2]) : 3072, K = argc > 3 ? atoll(argv[3]) : 768;
    if ((N & 7) || (K & 7)) { printf("N, K must be multiples of 8\n"); return 1; }
    void* lib = dlopen("vit_pytorch_amd/libvitk.so", RTLD_NOW);
    typedef int64_t (*splits_fn)(int64_t, int64_t, int64_t);
    typedef int (*tn_fn)(const void*, int64_t, const void*, int64_t, void*, int, int64_t, int, int64_t, int64_t, int64_t, float*, int64_t, void*);
    splits_fn vitk_splits = lib ? (splits_fn)dlsym(lib, "vitk_gemm_tn_splits") : nullptr;
    tn_fn vitk_tn = lib ? (tn_fn)dlsym(lib, "vitk_gemm_tn_bf16") : nullptr;
    if (!vitk_splits || !vitk_tn) { printf("libvitk.so not found (run from the repo root after building)\n"); return 1; }
    const long long splits = argc > 4 ? atoll(argv[4]) : vitk_splits(M, N, K);
    const int tiles_n = (int)((N + 255) / 256), tiles_k = (int)((K + 255) / 256), nwg = tiles_n * tiles_k;
    long long rps = (M + splits - 1) / splits; rps = (rps + 31) / 32 * 32;
    __bf16 *dY, *X; float *ws, *out_new, *out_ref;
    CK(hipMalloc(&dY, M * N * 2)); CK(hipMalloc(&X, M * K * 2)); CK(hipMalloc(&ws, splits * N * K * 4));
    CK(hipMalloc(&out_new, N * K * 4)); CK(hipMalloc(&out_ref, N * K * 4));
    const long long scratch_per_wave = (rps / 32 + 4) * 4 * 1024;
    char* scratch; CK(hipMalloc(&scratch, scratch_per_wave * 4 * nwg * splits));
    w_fill_kernel<<<(unsigned)((M * N + 255) / 256), 256>>>(dY, M * N, 1u, 0.02f);
    w_fill_kernel<<<(unsigned)((M * K + 255) / 256), 256>>>(X, M * K, 2u, 1.0f);
    struct Var { const char* name; kern_t k; int lds; };
    const Var vars[] = {
        {"4 stages padded: all", tn_w128_kernel<0, 4, 1>, 4 * Geo<1>::STAGE}, {"4 stages padded: no DMA", tn_w128_kernel<1, 4, 1>, 4 * Geo<1>::STAGE},
        {"4 stages padded: no reads", tn_w128_kernel<2, 4, 1>, 4 * Geo<1>::STAGE}, {"4 stages padded: MFMA only", tn_w128_kernel<3, 4, 1>, 4 * Geo<1>::STAGE},
        {"4 stages padded: DMA only", tn_w128_kernel<6, 4, 1>, 4 * Geo<1>::STAGE}, {"4 stages padded: reads only", tn_w128_kernel<5, 4, 1>, 4 * Geo<1>::STAGE},
        {"3 stages padded: all", tn_w128_kernel<0, 3, 1>, 3 * Geo<1>::STAGE}, {"3 stages padded: DMA only", tn_w128_kernel<6, 3, 1>, 3 * Geo<1>::STAGE},
        {"4 stages unpadded: all", tn_w128_kernel<0, 4, 0>, 4 * Geo<0>::STAGE}, {"4 stages unpadded: reads only", tn_w128_kernel<5, 4, 0>, 4 * Geo<0>::STAGE},
        {"5 stages unpadded: all", tn_w128_kernel<0, 5, 0>, 5 * Geo<0>::STAGE}, {"5 stages unpadded: DMA only", tn_w128_kernel<6, 5, 0>, 5 * Geo<0>::STAGE},
        {"5 stages unpadded: no DMA", tn_w128_kernel<1, 5, 0>, 5 * Geo<0>::STAGE},
        {"DMA in groups 0-7: all", tn_w128_kernel<0, 4, 1, 0, 1>, 4 * Geo<1>::STAGE},
        {"DMA in groups 8-15: all", tn_w128_kernel<0, 4, 1, 0, 2>, 4 * Geo<1>::STAGE},
        {"no s_setprio: all", tn_w128_kernel<0, 4, 1, 0, 3>, 4 * Geo<1>::STAGE},
        {"4 st. padded + 8 KiB stores/step, coupled", tn_w128_kernel<0, 4, 1, 1>, 4 * Geo<1>::STAGE},
        {"4 st. padded + 8 KiB stores/step, decoupled", tn_w128_kernel<0, 4, 1, 2>, 4 * Geo<1>::STAGE},
        {"4 st. padded, decoupled roles, DMA only", tn_w128_kernel<6, 4, 1, 2>, 4 * Geo<1>::STAGE},
    };
    const int nvars = (int)(sizeof(vars) / sizeof(vars[0]));
    for (int a = 0; a < nvars; ++a) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(vars[a].k), hipFuncAttributeMaxDynamicSharedMemorySize, vars[a].lds));
    auto run_kernel = [&](int a) {
        hipLaunchKernelGGL(vars[a].k, dim3((unsigned)(nwg * splits)), dim3(256), vars[a].lds, 0, dY, N, X, K, ws, (int)M, (int)N, (int)K, (int)rps, tiles_k, nwg, scratch, scratch_per_wave);
    };
    auto run_new = [&]() { run_kernel(0); w_reduce_kernel<<<(unsigned)((N * K + 255) / 256), 256>>>(ws, (int)splits, N * K, out_new); };
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
    const double fl = 2.0 * M * N * K;
    float ms_new = 0, ms_ref = 0, ms_k = 0;
    for (int rep = 0; rep < 2; ++rep) {        // interleaved A/B, second round reported
        CK(hipEventRecord(e0)); for (int i = 0; i < 20; ++i) run_new(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_new, e0, e1));
        CK(hipEventRecord(e0)); for (int i = 0; i < 20; ++i) run_ref(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_ref, e0, e1));
    }
    printf("w128 kernel + fold: %.3f ms = %.0f TF/s;  production vitk_gemm_tn_bf16: %.3f ms = %.0f TF/s\n", ms_new / 20, fl / (ms_new / 20) / 1e9,
           ms_ref / 20, fl / (ms_ref / 20) / 1e9);
    // every variant that computes the real thing is checked against the production result, then all are timed (kernel alone)
    const char* only = getenv("TN_PROBE_VARS");      // e.g. "0,4": time only these variants (profiling runs)
    for (int ab = 0; ab < nvars; ++ab) {
        if (only) { char key[8]; snprintf(key, sizeof key, ",%d,", ab); char buf[256]; snprintf(buf, sizeof buf, ",%s,", only); if (!strstr(buf, key)) continue; }
        const bool real = strstr(vars[ab].name, ": all") != nullptr || strstr(vars[ab].name, "coupled") != nullptr;
        double relv = -1;
        if (real) {
            CK(hipMemset(ws, 0xff, splits * N * K * 4));
            run_kernel(ab); w_reduce_kernel<<<(unsigned)((N * K + 255) / 256), 256>>>(ws, (int)splits, N * K, out_new);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(a.data(), out_new, N * K * 4, hipMemcpyDeviceToHost));
            double nu = 0, de = 0;
            for (size_t i = 0; i < a.size(); ++i) { nu += ((double)a[i] - b[i]) * ((double)a[i] - b[i]); de += (double)b[i] * b[i]; }
            relv = de > 0 ? sqrt(nu / de) : -1;
        }
        run_kernel(ab);
        CK(hipEventRecord(e0)); for (int i = 0; i < 20; ++i) run_kernel(ab); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_k, e0, e1));
        printf("  kernel alone, %-30s: %.1f us (%.0f TF/s equivalent)", vars[ab].name, ms_k / 20 * 1e3, fl / (ms_k / 20) / 1e9);
        if (real) printf("   rel. error %.2e %s", relv, relv >= 0 && relv < 1e-5 ? "OK" : "NUMERICS_DIFFERENT");
        printf("\n");
    }
    return 0;
}
