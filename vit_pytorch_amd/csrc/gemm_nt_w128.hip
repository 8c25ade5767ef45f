// gemm_nt_w128.hip -- persistent NT GEMM for gfx950, round 5:  C[M,N] = A[M,K] . W[N,K]^T (+ fused epilogue)
//
// nn.Linear forward (vit.py:20,23,44,47) and the dX GEMMs of its autograd, like gemm_nt_persist.hip -- the same 256 x 256 tiles in
// the same grouped, XCD-aware order, the same LDS image of a K-step (64-byte rows, chunk swizzle [0,2,3,1], W rows de-interleaved
// so that a lane's four B fragments are four consecutive output columns), same epilogues and therefore bit-identical results -- but
// built like the weight-gradient kernel of round 4 (gemm_tn_w128.hip):
//   * FOUR waves, one per SIMD, each owning a 128 x 128 wave tile: 16 fragment reads (ds_read_b128) per 64 MFMAs instead of 12 per
//     32, the 64 accumulator tiles pinned in 256 AGPRs by asm MFMAs ("+a"), TWO complete fragment sets in VGPRs so that every read
//     of K-step t + 1 is issued under the MFMAs of K-step t, one instruction per MFMA group, and none is waited for after its issue;
//   * operands by LDS-DMA through BUFFER DESCRIPTORS (buffer_load_dwordx4 ... lds): per-lane offsets formed once per kernel, a scalar
//     K-step offset, one descriptor per tile whose range ends with the tile (rows past M, or past a 128-row tile, arrive as zeros
//     without memory traffic); a K-step is 8 DMA instructions per wave, one per second MFMA group; one barrier per K-step;
//   * the K-step stream is continuous across tiles (the first four K-steps of tile n + 1 are in flight / landed during the
//     epilogue of tile n, its first fragments are read under the last MFMAs of tile n); the first K-step of a tile writes the
//     accumulators with C = 0 (no zeroing pass); the first two K-steps after an epilogue wait with an exact count that lets
//     that epilogue's stores stay in flight (vmcnt retires in order: the DMA pieces they need are older than the stores).
// Round 6 (AW = 1, the default): the ACTIVATION operand arrives as full 128-byte lines.  The feed is bound by cache-line REQUESTS, not bytes
// (DESIGN_HISTORY, round 2), and a 64-byte row piece of a row-major activation matrix is HALF a line whose other half is requested a K-step
// later: an instruction now fetches 8 rows x 128 bytes = the slices of TWO K-steps (8 full lines instead of 16 half lines), into a ring of
// three 32 KiB activation slots of 128-byte rows (chunk position s of LDS row R holds logical chunk s ^ ((R >> 1) & 7): conflict-free
// ds_read_b128), beside the four 16 KiB stages of the K-blocked W image (already contiguous KiB pieces).  Same pieces per wave and K-step
// (4 + 4), same counted waits, same fragments, same MFMA order: bit-identical results.  AW = 0 is round 5's feed (VITK_NTW_A128=0).
// The epilogue is the 8-wave kernel's, over two 64-column blocks per wave; operand rows it reads (16-bit residual, gelu' factor,
// f32 residual) come by uncounted asm loads a few fragment rows ahead -- one wave per SIMD has the registers for a deeper prefetch.
#include "common.h"
#include "gemm_nt_plan.h"
#include "gemm_nt_epi.h"
#include <stdlib.h>
#include <type_traits>

#ifdef VITK_HALF_IS_F16
#define NTW_MFMA_ASM "v_mfma_f32_16x16x32_f16"
#else
#define NTW_MFMA_ASM "v_mfma_f32_16x16x32_bf16"
#endif

namespace {

constexpr int V_TILE = 256 * 64;                    // one operand, one K-step: 256 rows of 64 bytes
constexpr int V_STAGE = 2 * V_TILE;                 // 32 KiB
constexpr int V_RING = 4 * V_STAGE;                 // 128 KiB
constexpr int V_LDS_MAX = V_RING + 32768;           // + bias image
// AW = 1: three activation slots (a K-step PAIR each: 256 rows of 128 bytes) + four W stages; the bias comes from global memory
constexpr int X_ASLOT = 256 * 128;                  // 32 KiB
constexpr int X_WBASE = 3 * X_ASLOT;                // 96 KiB
constexpr int X_LDS = X_WBASE + 4 * V_TILE;         // 160 KiB: the whole LDS of a CU

#define V_PIN() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ int v_swz(int x) { return (0x1320 >> (4 * (x & 3))) & 3; }   // permutation [0,2,3,1] (the 8-wave kernel's)

__device__ __forceinline__ void v_grouped_tile(int t, int tiles_m, int tiles_n, int gn, int& tm, int& tn) {
    const int per_group = gn * tiles_m;
    const int g = t / per_group;
    const int r = t - g * per_group;
    const int rem = tiles_n - g * gn;
    const int w = rem < gn ? rem : gn;
    tm = r / w;
    tn = g * gn + (r - tm * w);
}

struct NtwArgs {
    const char* A; long long lda;      // element strides; operands are 2-byte elements
    const char* W; long long ldw;      // ldw == 0: K-blocked (vitk_pack_w_nt)
    void* C; long long ldc;
    int M, N, K;
    const __bf16* bias; const void* resid; __bf16* aux; float* csum;
    int tiles_n, group_n, tiles_m, n_tiles, nt;      // FULL interior tiles only: rows [0, 256 tiles_m), N % 256 == 0
    int dbg;            // experiments: bit 0 = skip the epilogue (main loop alone), bit 1 = strict waits after an epilogue,
                        // bits 8..15 = late start of every second workgroup (x s_sleep 127)
};

template <int OFF> __device__ __forceinline__ bf16x8 v_rd(unsigned lds_addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(OFF) : "memory");
    return v;
}
#ifdef NTW_PROBE
// experiments (ABL bit 3, timing only, results are wrong): the same stream with v_mfma_f32_32x32x16 -- twice the flops per instruction
// and per operand register read -- on the fragments the 16 x 16 x 32 form reads; accumulators as 16 tuples of 16 registers
typedef float v_f32x16 __attribute__((ext_vector_type(16)));
#ifdef VITK_HALF_IS_F16
#define NTW_MFMA32_ASM "v_mfma_f32_32x32x16_f16"
#else
#define NTW_MFMA32_ASM "v_mfma_f32_32x32x16_bf16"
#endif
__device__ __forceinline__ void v_mfma32(v_f32x16& c, const bf16x8& a, const bf16x8& b) {
    asm volatile("" NTW_MFMA32_ASM " %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
#endif
// MFMA as asm with the accumulator pinned to AGPRs; the Z form writes A.B (C = 0): the first K-step of a tile
template <bool Z> __device__ __forceinline__ void v_mfma(f32x4& c, const bf16x8& a, const bf16x8& b) {
    if constexpr (Z) asm volatile("" NTW_MFMA_ASM " %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
    else asm volatile("" NTW_MFMA_ASM " %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

__device__ __forceinline__ void v_gload_bf16x4(bf16x4& d, const __bf16* p) { asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }

// ABL (experiments, tools/nt_probe.hip): bit 0 no LDS-DMA in the loop, bit 1 no fragment reads, bit 2 no MFMA
template <int EPI, int ABL, int AW>
__global__ __launch_bounds__(256) void gemm_ntw_kernel(const NtwArgs p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr bool F32OUT = (EPI == VITK_EPI_RESID);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- this workgroup's tiles (gemm_nt_persist.hip's static lists): XCD x owns a contiguous run of the main list and of the tail
    //      list; its workgroups take every L-th tile of the concatenation ----
    const int xcd = blockIdx.x & 7, l0 = blockIdx.x >> 3, L = gridDim.x >> 3;
    const int ms = (int)(((long long)xcd * p.n_tiles) >> 3), count = (int)(((long long)(xcd + 1) * p.n_tiles) >> 3) - ms;
    if (l0 >= count) return;
    // experiments (p.dbg bits 8..15): every second workgroup of an XCD starts that many s_sleep 127 (~3.4 us each) late, i.e. out of phase
    // with its neighbours -- do epilogue store bursts that fall into other workgroups' main loops pay for the idle start?
    if ((p.dbg >> 8) && (l0 & 1)) { for (int i = (p.dbg >> 8) & 0xff; i > 0; --i) __builtin_amdgcn_s_sleep(127); }
    auto decode = [&](int idx, int& m0, int& n0, int& mt) {
        int tn;
        v_grouped_tile(ms + idx, p.tiles_m, p.tiles_n, p.group_n, mt, tn);
        m0 = mt * 256; n0 = tn * 256;
    };

    // ---- producer: the LDS-DMA stream runs four K-steps ahead of the MFMAs, across tile boundaries ----
    // 64-byte rows, a wave instruction fills 16 rows; wave w owns pieces 4w .. 4w + 3 of each operand tile.  The LDS image is lane-linear, so
    // the bank swizzle sits in the SOURCE offset: position s of LDS row R holds logical 16-byte chunk s ^ v_swz(R >> 2).  Activation tile:
    // LDS row R = tile row R.  W tile: LDS row R = 64 q + 16 fn + c holds W row 64 q + 4 c + fn (a lane's four B fragments of a 64-column
    // block are then 4 consecutive output columns); K-blocked W (ldw == 0) IS that image, block (n-tile, K-step) after block.
    // AW = 1, activation tile: an instruction fills 8 LDS rows of 128 bytes (a K-step pair); wave w owns pieces 8w .. 8w + 7 of a slot, piece i =
    // rows 64 w + 8 i + (lane >> 3).  LDS row R, position s holds logical chunk s ^ ((R >> 1) & 7) = s ^ (4 (i & 1) + (lane >> 4)): one lane
    // offset for even pieces, one for odd ones; the row offset of piece i (8 i rows) rides in the scalar offset.
    const int srow = lane >> 2, spos = lane & 3;
    const int schunk = spos ^ v_swz(lane >> 4);
    int avo[4], wvo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if constexpr (AW) avo[j] = (int)(((long long)(64 * wave + (lane >> 3)) * p.lda + (((lane & 7) ^ (4 * (j & 1) + (lane >> 4))) * 8)) * 2);      // j & 1 = piece parity
        else avo[j] = (int)(((long long)(64 * wave + 16 * j + srow) * p.lda + schunk * 8) * 2);
        wvo[j] = p.ldw == 0 ? (4 * wave + j) * 1024 + lane * 16 : (int)(((long long)(64 * wave + 4 * srow + j) * p.ldw + schunk * 8) * 2);
    }
    const int a_rows8 = (int)(p.lda * 16);              // AW = 1: bytes between the rows of consecutive pieces (8 rows)
    const int w_kstride = p.ldw == 0 ? V_TILE : 64;     // bytes between consecutive K-steps
    __amdgpu_buffer_rsrc_t a_rs, w_rs;
    int a_so = 0, w_so = 0;                             // scalar byte offsets of the producer's K-step
    // (AW = 1: the two operands change tile at different K-steps -- the activation stream runs up to three K-step PAIRS ahead, the W stream
    //  four K-steps -- hence one setup per operand; past the end of the tile list the descriptor has range 0)
    auto setup_a = [&](int idx) __attribute__((always_inline)) {
        if (idx < count) {
            int m0, n0, mt;
            decode(idx, m0, n0, mt);
            const long long abytes = (255LL * p.lda + p.K) * 2;       // up to the last element of the tile's last row
            a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (long long)m0 * p.lda * 2), 0, (int)abytes, 0x00020000);
        } else a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0, 0x00020000);
        a_so = 0;
    };
    auto setup_w = [&](int idx) __attribute__((always_inline)) {
        if (idx < count) {
            int m0, n0, mt;
            decode(idx, m0, n0, mt);
            if (p.ldw == 0) {
                w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (long long)(n0 >> 8) * p.nt * V_TILE), 0, p.nt * V_TILE, 0x00020000);
            } else {
                const long long wbytes = (255LL * p.ldw + p.K) * 2;
                w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (long long)n0 * p.ldw * 2), 0, (int)wbytes, 0x00020000);
            }
        } else w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0, 0x00020000);
        w_so = 0;
    };
    auto setup_src = [&](int idx) { setup_a(idx); setup_w(idx); };
    // piece q of the producer's K-step into stage `stg`: q < 4 activation piece 4w + q, else W piece 4w + q - 4
    auto dma = [&](int stg, int q) __attribute__((always_inline)) {
        char* dst = lds + stg * V_STAGE + (q >= 4 ? V_TILE : 0) + (wave * 4 + (q & 3)) * 1024;
        if (q < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (void __attribute__((address_space(3)))*)dst, 16, avo[q & 3], a_so, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (void __attribute__((address_space(3)))*)dst, 16, wvo[q & 3], w_so, 0, 0);
    };
    // after the 8 pieces of a K-step.  The producer changes tile at a FIXED point of the consumer's tile (before its K-step nt - 4: the stream
    // runs four K-steps ahead), so no K-step carries a tile-switch test.  Past the end of the tile list the descriptors have range 0: the
    // pieces still count (the waits stay uniform) but fetch nothing and write zeros into stages nobody reads.
    auto advance = [&]() __attribute__((always_inline)) { a_so += 64; w_so += w_kstride; };
    auto next_src = [&](int idx) __attribute__((always_inline)) { setup_src(idx); };
    // AW = 1: activation piece i (0..7) of the producer's K-step pair into slot `slot`; W piece j (0..3) of the producer's K-step into stage `stg`
    auto dma_a = [&](int slot, int i) __attribute__((always_inline)) {
        char* dst = lds + slot * X_ASLOT + (wave * 8 + i) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (void __attribute__((address_space(3)))*)dst, 16, avo[i & 1], a_so + i * a_rows8, 0, 0);
    };
    auto dma_w = [&](int stg, int j) __attribute__((always_inline)) {
        char* dst = lds + X_WBASE + stg * V_TILE + (wave * 4 + j) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (void __attribute__((address_space(3)))*)dst, 16, wvo[j], w_so, 0, 0);
    };

    // ---- bias -> LDS (above the ring), once ----
    const char* bias_lds = lds + V_RING;
    if constexpr (q_has_bias<EPI>() && !AW) {
        const int ncols = p.tiles_n * 256;
        for (int i = tid * 8; i < ncols; i += 256 * 8) {
            bf16x8 v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (p.bias && i < p.N) v = *reinterpret_cast<const bf16x8*>(p.bias + i);
            *reinterpret_cast<bf16x8*>(lds + V_RING + i * 2) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // written before the prologue's barrier publishes it
    }

    // ---- consumer ----
    const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)lds);
    unsigned a_rd, w_rd;                                 // + f * 1024 / + fn * 1024, + stage  (AW = 1: + f * 2048, + slot; first K-step of a pair)
    unsigned a_rd1 = 0;                                  // AW = 1: the second K-step of a pair (logical chunks 4..7: position ^ 4)
    {
        const int fi = lane & 15, fg = lane >> 4;
        const int fpos = fg ^ v_swz(fi >> 2);
        if constexpr (AW) {
            const int xpos = fg ^ ((fi >> 1) & 7);
            a_rd = lds_base + (wm * 128 + fi) * 128 + xpos * 16;
            a_rd1 = lds_base + (wm * 128 + fi) * 128 + (xpos ^ 4) * 16;
            w_rd = lds_base + X_WBASE + (wn * 128 + fi) * 64 + fpos * 16;
        } else {
            a_rd = lds_base + (wm * 128 + fi) * 64 + fpos * 16;
            w_rd = lds_base + V_TILE + (wn * 128 + fi) * 64 + fpos * 16;
        }
    }
    constexpr int A_FRAG = AW ? 2048 : 1024;             // bytes between the activation fragments of a wave (16 rows)
    int sp = 0;                 // AW = 1: K-step pair counter mod 3: the slot of the pair the MFMAs are in

    int stg = 0;                // K-step counter mod 4: the stage whose fragments are in registers (the DMA of this K-step refills it)
    bf16x8 xa[8], wa[8], xb[8], wb[8];
    f32x4 acc[8][8];            // acc[fn][f][j]: row 16 f + 4 fg + j, column 64 (fn >> 2) + 4 fi + (fn & 3) of the wave tile
#ifdef NTW_PROBE
    v_f32x16 acc32[16];
    if constexpr (ABL & 8) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc32[i] = v_f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    }
#endif

    // one group of a K-step: 4 MFMAs on the current fragments (activation fragment fm x W fragments 4h .. 4h + 3), one fragment of the next
    // K-step, every second group one DMA piece.  [measured, tools/nt_probe, FF1 shape: LDS-DMA alone 112 us, MFMA alone 111 us, both 160-166 us:
    // the wave's matrix pipe runs dry while a DMA instruction waits to issue.  NOT the waves queueing behind one another at the texture-address
    // unit: with the four waves' pieces at four different MFMA slots (one piece per two slots CU-wide) the loop ran 165.9 vs 164.9 us.  NOR
    // the lone wave of a SIMD being stuck at the issue: the same stream on EIGHT waves (two per SIMD, 128 x 64 wave tiles, the partner's MFMAs
    // in the gaps) ran 169 vs 165 us, its epilogues level too (profiles/r05c_nt_probe_8wave_vs_4wave.log; the flavour was not kept).  What is
    // left is the chip: matrix cores and the LDS-DMA feed at full rate together draw more than either alone (DVFS, MI355X_MICROARCH.md).]
#ifdef NTW_PROBE
#define V_MFMA32(G, I_, XC, WC) do { if constexpr (!(ABL & 4) && (ABL & 8)) v_mfma32(acc32[(2 * (G) + (I_)) & 15], XC[(G) >> 1], WC[((G) & 1) * 4 + (I_)]); } while (0)
#else
#define V_MFMA32(G, I_, XC, WC) do { } while (0)
#endif
#define V_GROUP(G, Z, XC, WC, XN, WN) do { \
        constexpr int fm_ = (G) >> 1, h_ = ((G) & 1) * 4; \
        if constexpr (!(ABL & 4) && !(ABL & 8)) v_mfma<Z>(acc[h_ + 0][fm_], XC[fm_], WC[h_ + 0]); \
        V_MFMA32(G, 0, XC, WC); \
        if constexpr (!(ABL & 2)) { if constexpr ((G) < 8) XN[(G)] = v_rd<(G) * A_FRAG>(rdA); else WN[(G) - 8] = v_rd<((G) - 8) * 1024>(rdW); } \
        if constexpr (!(ABL & 4) && !(ABL & 8)) v_mfma<Z>(acc[h_ + 1][fm_], XC[fm_], WC[h_ + 1]); \
        if constexpr (!(ABL & 4) && !(ABL & 8)) v_mfma<Z>(acc[h_ + 2][fm_], XC[fm_], WC[h_ + 2]); \
        if constexpr (!(ABL & 1) && ((G) & 1)) { V_PIN(); if constexpr (AW) { if constexpr ((G) < 8) dma_a(dslot_, di0_ + ((G) >> 1)); else dma_w(stg, ((G) >> 1) - 4); } else dma(stg, (G) >> 1); V_PIN(); } \
        V_MFMA32(G, 1, XC, WC); \
        if constexpr (!(ABL & 4) && !(ABL & 8)) v_mfma<Z>(acc[h_ + 3][fm_], XC[fm_], WC[h_ + 3]); \
    } while (0)
    // one K-step.  WAIT = the counted vmcnt statement: own pieces of K-step t + 2 landed (t + 3, t + 4 fly)
    // AW = 1, H = the K-step's place in its pair P.  H = 0: reads the pair's second K-step (slot sp, positions ^ 4), issues the second half of
    // pair P + 2 into slot sp + 2; H = 1: reads the first K-step of pair P + 1 (slot sp + 1), issues the first half of pair P + 3 into slot sp
    // (read by everyone as of the previous barrier).  Either way the W pieces of K-step t + 4 go into stage stg.
#define V_STEP(Z, XC, WC, XN, WN, WAIT, H) do { \
        const unsigned soff_ = ((stg + 1) & 3) * (AW ? V_TILE : V_STAGE); \
        const int sp1_ = sp == 2 ? 0 : sp + 1; \
        const int dslot_ = (H) ? sp : (sp == 0 ? 2 : sp - 1); \
        constexpr int di0_ = (H) ? 0 : 4; \
        const unsigned rdA = AW ? ((H) ? a_rd + sp1_ * X_ASLOT : a_rd1 + sp * X_ASLOT) : a_rd + soff_; \
        const unsigned rdW = w_rd + soff_; \
        __builtin_amdgcn_s_setprio(1); \
        V_GROUP(0, Z, XC, WC, XN, WN); V_GROUP(1, Z, XC, WC, XN, WN); V_GROUP(2, Z, XC, WC, XN, WN); V_GROUP(3, Z, XC, WC, XN, WN); \
        V_GROUP(4, Z, XC, WC, XN, WN); V_GROUP(5, Z, XC, WC, XN, WN); V_GROUP(6, Z, XC, WC, XN, WN); V_GROUP(7, Z, XC, WC, XN, WN); \
        V_GROUP(8, Z, XC, WC, XN, WN); V_GROUP(9, Z, XC, WC, XN, WN); V_GROUP(10, Z, XC, WC, XN, WN); V_GROUP(11, Z, XC, WC, XN, WN); \
        V_GROUP(12, Z, XC, WC, XN, WN); V_GROUP(13, Z, XC, WC, XN, WN); V_GROUP(14, Z, XC, WC, XN, WN); V_GROUP(15, Z, XC, WC, XN, WN); \
        __builtin_amdgcn_s_setprio(0); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      /* the next K-step's fragments are in registers */ \
        if constexpr (!(ABL & 1)) { WAIT; } \
        V_PIN(); \
        __builtin_amdgcn_s_barrier();           /* stage t + 2 visible to all, stage t + 1 read by all */ \
        V_PIN(); \
        if constexpr (!(ABL & 1)) { if constexpr (AW) { w_so += w_kstride; if constexpr (!(H)) a_so += 128; } else advance(); } \
        if constexpr (AW && (H)) sp = sp1_; \
        stg = (stg + 1) & 3; \
    } while (0)
#define V_WAIT16 asm volatile("s_waitcnt vmcnt(16)" ::: "memory")
    // ---- epilogue of one tile: registers -> global, full lines, stores not waited for ----
    // the 8-wave kernel's, over the wave's two 64-column blocks: "row" r = 2 f + qq covers fragment row f (4 output rows per lane) of
    // block qq.  Every tile is interior (the launch takes full tiles only), so operand rows come by uncounted asm loads D rows ahead
    // and are waited for by exact counts (gemm_nt_epi.h).
    bf16x4 bq[2] = {bf16x4{0, 0, 0, 0}, bf16x4{0, 0, 0, 0}};      // AW = 1: the lane's bias values of the tile, loaded at the tile's first K-step
    auto epilogue = [&](int m0, int n0, int mt) __attribute__((always_inline)) {
        constexpr int NR = 16;
#ifdef NTW_PROBE
        if constexpr (ABL & 8) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("" :: "a"(acc32[i]));
            return;
        }
#endif
        if (p.dbg & 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("" :: "a"(acc[i][j]));
            return;
        }
        // lane / wave coordinates re-derived behind an opaque statement: hipcc otherwise hoists the epilogue's per-row index arithmetic
        // above the tile loop and carries it through the main loop in scratch
        int elane;              // the lane id, not carried through the main loop (volatile: hipcc hoists the builtin out of the tile loop)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(elane));
        int fi = elane & 15, fg = elane >> 4, ewm = wm, ewn = wn;
        asm volatile("" : "+v"(fi), "+v"(fg), "+s"(ewm), "+s"(ewn));
        const int mrow0 = m0 + ewm * 128 + 4 * fg;              // + 16 f + j
        const int ncolw = n0 + ewn * 128;                       // first column of the wave tile
        const long long obase4 = (long long)mrow0 * p.ldc + ncolw + 4 * fi;      // element (row mrow0, the lane's 4 columns of block 0)
        f32x4 b4[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        if constexpr (q_has_bias<EPI>()) {
            if constexpr (AW) asm volatile("" : "+v"(bq[0]), "+v"(bq[1]));       // (asm loads of ~nt K-steps ago: every counted wait since covered them)
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                bf16x4 bb;
                if constexpr (AW) bb = bq[qq];
                else bb = *reinterpret_cast<const bf16x4*>(bias_lds + (ncolw + qq * 64 + 4 * fi) * 2);
                b4[qq] = f32x4{(float)bb[0], (float)bb[1], (float)bb[2], (float)bb[3]};
            }
        }
        if constexpr (F32OUT) {
            // lane: rows mrow0 + 16 f + j, 4 consecutive f32 columns: 16 lanes = 256 contiguous bytes of a row
            float* Cf = reinterpret_cast<float*>(p.C);
            const float* Rf = reinterpret_cast<const float*>(p.resid);
            constexpr int D = 4;
            f32x4 r[D][4];
            auto fetch = [&](int rr_, f32x4 (&dst)[4]) __attribute__((always_inline)) {
                const int f = rr_ >> 1, qq = rr_ & 1;
                long long orow = obase4 + (long long)(f * 16) * p.ldc + qq * 64;
                asm volatile("" : "+v"(orow));
                const float* rp = Rf + orow;
#pragma unroll
                for (int j = 0; j < 4; ++j) q_gload_f32x4(dst[j], rp + (long long)j * p.ldc);
            };
#pragma unroll
            for (int i = 0; i < D; ++i) fetch(i, r[i]);
            auto row = [&](auto rc) __attribute__((always_inline)) {
                constexpr int R_ = decltype(rc)::value;
                constexpr int f = R_ >> 1, qq = R_ & 1;
                // hipcc places the AGPR -> VGPR copy of an asm output right behind its DEFINITION (here: the last MFMAs, inside their latency,
                // all 256 of them live through the epilogue, ~90 registers of it in scratch): re-define the row's accumulators here
                V_PIN();
                asm volatile("" : "+a"(acc[4 * qq + 0][f]), "+a"(acc[4 * qq + 1][f]), "+a"(acc[4 * qq + 2][f]), "+a"(acc[4 * qq + 3][f]));
                f32x4 (&rr)[4] = r[R_ % D];
                q_wait_regs4<q_epi_younger(R_, NR, D, 4)>(rr[0], rr[1], rr[2], rr[3]);
                long long ocp = obase4 + (long long)(f * 16) * p.ldc + qq * 64;
                asm volatile("" : "+v"(ocp));
                float* cp = Cf + ocp;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 v = f32x4{acc[4 * qq + 0][f][j], acc[4 * qq + 1][f][j], acc[4 * qq + 2][f][j], acc[4 * qq + 3][f][j]} + b4[qq];
                    v += rr[j];
                    *reinterpret_cast<f32x4*>(cp + (long long)j * p.ldc) = v;
                }
                if constexpr (R_ + D < NR) fetch(R_ + D, rr);
            };
            row(std::integral_constant<int, 0>{}); row(std::integral_constant<int, 1>{}); row(std::integral_constant<int, 2>{}); row(std::integral_constant<int, 3>{});
            row(std::integral_constant<int, 4>{}); row(std::integral_constant<int, 5>{}); row(std::integral_constant<int, 6>{}); row(std::integral_constant<int, 7>{});
            row(std::integral_constant<int, 8>{}); row(std::integral_constant<int, 9>{}); row(std::integral_constant<int, 10>{}); row(std::integral_constant<int, 11>{});
            row(std::integral_constant<int, 12>{}); row(std::integral_constant<int, 13>{}); row(std::integral_constant<int, 14>{}); row(std::integral_constant<int, 15>{});
        } else if constexpr (EPI == VITK_EPI_RESID16) {
            // the residual epilogue with the stream in the 16-bit type: lane -> (row, 4 columns), 8-byte loads and stores (16 lanes = 128
            // contiguous bytes of a row); the sum is formed in f32 and rounded once
            __bf16* Cb = reinterpret_cast<__bf16*>(p.C);
            const __bf16* Rb = reinterpret_cast<const __bf16*>(p.resid);
            constexpr int D = 6;
            bf16x4 r[D][4];
            auto fetch = [&](int rr_, bf16x4 (&dst)[4]) __attribute__((always_inline)) {
                const int f = rr_ >> 1, qq = rr_ & 1;
                long long orow = obase4 + (long long)(f * 16) * p.ldc + qq * 64;
                asm volatile("" : "+v"(orow));
                const __bf16* rp = Rb + orow;
#pragma unroll
                for (int j = 0; j < 4; ++j) v_gload_bf16x4(dst[j], rp + (long long)j * p.ldc);
            };
#pragma unroll
            for (int i = 0; i < D; ++i) fetch(i, r[i]);
            auto row = [&](auto rc) __attribute__((always_inline)) {
                constexpr int R_ = decltype(rc)::value;
                constexpr int f = R_ >> 1, qq = R_ & 1;
                // hipcc places the AGPR -> VGPR copy of an asm output right behind its DEFINITION (here: the last MFMAs, inside their latency,
                // all 256 of them live through the epilogue, ~90 registers of it in scratch): re-define the row's accumulators here
                V_PIN();
                asm volatile("" : "+a"(acc[4 * qq + 0][f]), "+a"(acc[4 * qq + 1][f]), "+a"(acc[4 * qq + 2][f]), "+a"(acc[4 * qq + 3][f]));
                bf16x4 (&rr)[4] = r[R_ % D];
                q_wait_regs4<q_epi_younger(R_, NR, D, 4)>(rr[0], rr[1], rr[2], rr[3]);
                long long ocp = obase4 + (long long)(f * 16) * p.ldc + qq * 64;
                asm volatile("" : "+v"(ocp));
                __bf16* cp = Cb + ocp;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 v = f32x4{acc[4 * qq + 0][f][j], acc[4 * qq + 1][f][j], acc[4 * qq + 2][f][j], acc[4 * qq + 3][f][j]} + b4[qq];
                    const bf16x4 q4 = rr[j];
                    v += f32x4{(float)q4[0], (float)q4[1], (float)q4[2], (float)q4[3]};
                    store4<__bf16>(cp + (long long)j * p.ldc, v);
                }
                if constexpr (R_ + D < NR) fetch(R_ + D, rr);
            };
            row(std::integral_constant<int, 0>{}); row(std::integral_constant<int, 1>{}); row(std::integral_constant<int, 2>{}); row(std::integral_constant<int, 3>{});
            row(std::integral_constant<int, 4>{}); row(std::integral_constant<int, 5>{}); row(std::integral_constant<int, 6>{}); row(std::integral_constant<int, 7>{});
            row(std::integral_constant<int, 8>{}); row(std::integral_constant<int, 9>{}); row(std::integral_constant<int, 10>{}); row(std::integral_constant<int, 11>{});
            row(std::integral_constant<int, 12>{}); row(std::integral_constant<int, 13>{}); row(std::integral_constant<int, 14>{}); row(std::integral_constant<int, 15>{});
        } else {
            // after the pair exchange: even lanes own row r = mrow0 + 16 f + 2 pr, odd lanes row r + 1, columns ncol8 .. + 7 of the block
            const int odd = fi & 1;
            __bf16* Cb = reinterpret_cast<__bf16*>(p.C);
            const long long obase = (long long)(mrow0 + odd) * p.ldc + ncolw + 8 * (fi >> 1);      // + 16 f ldc + 64 qq + 2 pr ldc
            constexpr bool AUX_IN = q_aux_in<EPI>();      // an (M, N) operand read in the epilogue (16-bit; MUL_AUX8: 8-bit codes)
            constexpr bool AUX8 = (EPI == VITK_EPI_MUL_AUX8);
            // [measured, round 6, one box, two interleaved runs (profiles/r06o_*): the codes of 4 / 8 / 16 fragment rows in flight (16 = the whole tile
            //  requested before the first store, so that no wait for a load waits for a store's acknowledge): dFF1 248-250 / 250-252 / 258-260 us;
            //  the 16-bit residual rows of EPI_RESID16 with 6 / 8 / 12 / 14 rows in flight: level.  4 and 6 stay.]
            constexpr int DP = AUX8 ? 4 : 6;
            using HPre = std::conditional_t<AUX8, q_u32x2, bf16x8>;
            HPre hpre[DP][2];
            auto fetch_pre = [&](int rr_, HPre (&dst)[2]) __attribute__((always_inline)) {
                const int f = rr_ >> 1, qq = rr_ & 1;
                long long oa = obase + (long long)(f * 16) * p.ldc + qq * 64;
                asm volatile("" : "+v"(oa));        // (opaque: hipcc otherwise forms the addresses of all 16 rows at the top of the epilogue -- 256 VGPRs)
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    if constexpr (AUX8) q_gload_u32x2(dst[pr], reinterpret_cast<const unsigned char*>(p.aux) + oa + (long long)(2 * pr) * p.ldc);
                    else q_gload_bf16x8(dst[pr], p.aux + oa + (long long)(2 * pr) * p.ldc);
                }
            };
            if constexpr (AUX_IN) {
#pragma unroll
                for (int i = 0; i < DP; ++i) fetch_pre(i, hpre[i]);
            }
            float cs[2][8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { cs[0][e] = 0.f; cs[1][e] = 0.f; }
            auto frow = [&](auto rc) __attribute__((always_inline)) {
                constexpr int R_ = decltype(rc)::value;
                constexpr int f = R_ >> 1, qq = R_ & 1;
                // hipcc places the AGPR -> VGPR copy of an asm output right behind its DEFINITION (here: the last MFMAs, inside their latency,
                // all 256 of them live through the epilogue, ~90 registers of it in scratch): re-define the row's accumulators here
                V_PIN();
                asm volatile("" : "+a"(acc[4 * qq + 0][f]), "+a"(acc[4 * qq + 1][f]), "+a"(acc[4 * qq + 2][f]), "+a"(acc[4 * qq + 3][f]));
                if constexpr (AUX_IN) q_wait_regs2<q_epi_younger(R_, NR, DP, 2)>(hpre[R_ % DP][0], hpre[R_ % DP][1]);
                long long o0 = obase + (long long)(f * 16) * p.ldc + qq * 64;
                asm volatile("" : "+v"(o0));
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    // rows j0 = 2 pr and j0 + 1 of this lane's 4 columns, rounded to the 16-bit type
                    const int j0 = 2 * pr;
                    const f32x4 bq = b4[qq];
                    const unsigned a0 = q_pack2b<q_has_bias<EPI>()>(acc[4 * qq + 0][f][j0], acc[4 * qq + 1][f][j0], bq[0], bq[1]);
                    const unsigned a1 = q_pack2b<q_has_bias<EPI>()>(acc[4 * qq + 2][f][j0], acc[4 * qq + 3][f][j0], bq[2], bq[3]);
                    const unsigned c0 = q_pack2b<q_has_bias<EPI>()>(acc[4 * qq + 0][f][j0 + 1], acc[4 * qq + 1][f][j0 + 1], bq[0], bq[1]);
                    const unsigned c1 = q_pack2b<q_has_bias<EPI>()>(acc[4 * qq + 2][f][j0 + 1], acc[4 * qq + 3][f][j0 + 1], bq[2], bq[3]);
                    // even lane keeps row j0 and receives the neighbour's 4 columns of it; odd lane likewise for row j0 + 1
                    const unsigned r0 = q_dpp_xor1(odd ? a0 : c0), r1 = q_dpp_xor1(odd ? a1 : c1);
                    const unsigned k0 = odd ? c0 : a0, k1 = odd ? c1 : a1;
                    const q_u32x4 w4 = odd ? q_u32x4{r0, r1, k0, k1} : q_u32x4{k0, k1, r0, r1};
                    const bf16x8 v = __builtin_bit_cast(bf16x8, w4);
                    const long long o = o0 + (long long)j0 * p.ldc;
                    if constexpr (EPI == VITK_EPI_NONE || EPI == VITK_EPI_BIAS) {
                        *reinterpret_cast<bf16x8*>(Cb + o) = v;
                    } else if constexpr (EPI == VITK_EPI_BIAS_GELU) {
                        *reinterpret_cast<bf16x8*>(p.aux + o) = v;
                        *reinterpret_cast<bf16x8*>(Cb + o) = q_narrow8(q_gelu8(q_widen8(v)));     // of the ROUNDED pre-activation
                    } else if constexpr (EPI == VITK_EPI_BIAS_GELU_DG || EPI == VITK_EPI_BIAS_GELU_DG8) {
                        q_f32x8 gl, dgl;
#ifdef NTW_PROBE
                        // experiments (timing only): p.dbg bit 5 = no GELU arithmetic (the stores alone), bit 6 = no stores (the arithmetic alone)
                        if (p.dbg & 32) { gl = q_widen8(v); dgl = gl; } else
#endif
                        q_gelu_both8(q_widen8(v), gl, dgl);              // of the ROUNDED pre-activation, like BIAS_GELU
#ifdef NTW_PROBE
                        if (p.dbg & 64) { const bf16x8 t0 = q_narrow8(gl); const q_u32x2 t1 = q_dg_encode8(dgl); asm volatile("" :: "v"(t0), "v"(t1)); continue; }
#endif
                        if constexpr (EPI == VITK_EPI_BIAS_GELU_DG8) *reinterpret_cast<q_u32x2*>(reinterpret_cast<unsigned char*>(p.aux) + o) = q_dg_encode8(dgl);
                        else *reinterpret_cast<bf16x8*>(p.aux + o) = q_narrow8(dgl);
                        *reinterpret_cast<bf16x8*>(Cb + o) = q_narrow8(gl);
                    } else if constexpr (AUX_IN) {
                        q_f32x8 fac;
                        if constexpr (AUX8) fac = q_dg_decode8(hpre[R_ % DP][pr]);
                        else if constexpr (EPI == VITK_EPI_MUL_AUX) fac = q_widen8(hpre[R_ % DP][pr]);
                        else fac = q_gelu_grad8(q_widen8(hpre[R_ % DP][pr]));
                        const q_f32x8 g = q_widen8(v) * fac;
                        const bf16x8 g8 = q_narrow8(g);
#ifdef NTW_PROBE
                        if (p.dbg & 64) asm volatile("" :: "v"(g8)); else       // experiments (timing only): bit 6 = no stores
#endif
                        *reinterpret_cast<bf16x8*>(Cb + o) = g8;
                        q_cs_add8(cs[qq], g8);                          // of the ROUNDED values: what colsum(C) would read
                    }
                }
                if constexpr (AUX_IN) {
                    if constexpr (R_ + DP < NR) fetch_pre(R_ + DP, hpre[R_ % DP]);
                }
            };
            frow(std::integral_constant<int, 0>{}); frow(std::integral_constant<int, 1>{}); frow(std::integral_constant<int, 2>{}); frow(std::integral_constant<int, 3>{});
            frow(std::integral_constant<int, 4>{}); frow(std::integral_constant<int, 5>{}); frow(std::integral_constant<int, 6>{}); frow(std::integral_constant<int, 7>{});
            frow(std::integral_constant<int, 8>{}); frow(std::integral_constant<int, 9>{}); frow(std::integral_constant<int, 10>{}); frow(std::integral_constant<int, 11>{});
            frow(std::integral_constant<int, 12>{}); frow(std::integral_constant<int, 13>{}); frow(std::integral_constant<int, 14>{}); frow(std::integral_constant<int, 15>{});
            if constexpr (AUX_IN) {
                if (p.csum) {
                    // bias gradient by-product: the 8 lanes (c ^ 1, 4 row groups) that own the same 8 columns are summed in
                    // registers; one partial row per (m-tile, wm)
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            float v = cs[qq][e];
                            v += __builtin_bit_cast(float, q_dpp_xor1(__builtin_bit_cast(unsigned, v)));
                            unsigned u = __builtin_bit_cast(unsigned, v);
                            auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
                            v = __builtin_bit_cast(float, (unsigned)a[0]) + __builtin_bit_cast(float, (unsigned)a[1]);
                            u = __builtin_bit_cast(unsigned, v);
                            auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                            cs[qq][e] = __builtin_bit_cast(float, (unsigned)b[0]) + __builtin_bit_cast(float, (unsigned)b[1]);
                        }
                        if (fg == 0 && !odd) {
                            float* cp = p.csum + (long long)(2 * mt + ewm) * p.N + ncolw + qq * 64 + 8 * (fi >> 1);
                            *reinterpret_cast<f32x4*>(cp) = f32x4{cs[qq][0], cs[qq][1], cs[qq][2], cs[qq][3]};
                            *reinterpret_cast<f32x4*>(cp + 4) = f32x4{cs[qq][4], cs[qq][5], cs[qq][6], cs[qq][7]};
                        }
                    }
                }
            }
        }
    };

    // ---- prologue: K-steps 0..3 in flight, 0 and 1 landed, the fragments of K-step 0 in registers ----
    setup_src(l0);
    if constexpr (!(ABL & 1)) {
        if constexpr (AW) {
            // pairs 0, 1 and the first half of pair 2; W K-steps 0..3.  In issue order: pair 0, W 0, W 1 | pair 1, W 2, W 3 | half of pair 2
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
#pragma unroll
                for (int i = 0; i < 8; ++i) dma_a(pr, i);
                a_so += 128;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) dma_w(2 * pr + k, j);
                    w_so += w_kstride;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) dma_a(2, i);
            asm volatile("s_waitcnt vmcnt(20)" ::: "memory");       // pair 0, W 0, W 1 landed
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int q = 0; q < 8; ++q) dma(s, q);
                advance();
            }
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        }
    }
    V_PIN();
    __builtin_amdgcn_s_barrier();           // also publishes the bias image
    V_PIN();
    {
        const unsigned rdA = a_rd, rdW = w_rd;
        xa[0] = v_rd<0 * A_FRAG>(rdA); xa[1] = v_rd<1 * A_FRAG>(rdA); xa[2] = v_rd<2 * A_FRAG>(rdA); xa[3] = v_rd<3 * A_FRAG>(rdA);
        xa[4] = v_rd<4 * A_FRAG>(rdA); xa[5] = v_rd<5 * A_FRAG>(rdA); xa[6] = v_rd<6 * A_FRAG>(rdA); xa[7] = v_rd<7 * A_FRAG>(rdA);
        wa[0] = v_rd<0 * 1024>(rdW); wa[1] = v_rd<1 * 1024>(rdW); wa[2] = v_rd<2 * 1024>(rdW); wa[3] = v_rd<3 * 1024>(rdW);
        wa[4] = v_rd<4 * 1024>(rdW); wa[5] = v_rd<5 * 1024>(rdW); wa[6] = v_rd<6 * 1024>(rdW); wa[7] = v_rd<7 * 1024>(rdW);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    V_PIN();
    __builtin_amdgcn_s_barrier();           // stage 0 has been read by everyone: K-step 0 may refill it
    V_PIN();
    if constexpr (ABL & 2) {
#pragma unroll
        for (int f = 0; f < 8; ++f) { xb[f] = xa[f]; wb[f] = wa[f]; }
    }

    // the first two K-steps after an epilogue: its stores may stay in flight behind the two K-steps' worth of DMA pieces the wait is
    // about (vmcnt retires in order and counts to 63); the first tile has nothing but DMA pieces in flight
    constexpr int ST_ROW = (F32OUT || EPI == VITK_EPI_RESID16 || q_two_outputs<EPI>()) ? 4 : 2;   // stores per epilogue row
    constexpr int VM_RELAX = 16 + 16 * ST_ROW > 63 ? 63 : 16 + 16 * ST_ROW;
    // What the exact count relies on: this wave's vector-memory operations retire IN ISSUE ORDER, loads and stores alike, so the DMA pieces the
    // wait is about (older than the epilogue's stores) have landed once at most VM_RELAX younger operations are outstanding.  That is how gfx950's
    // single vmcnt behaves for buffer / global operations of one wave (FLAT operations, which may resolve to LDS or scratch, are the documented
    // out-of-order case; none is issued here) -- and it is the SAME property every counted wait of this file and of gemm_nt_epi.h already rests on
    // (a residual row's exact-count wait passes stores issued before it).  Held empirically by the bit-identity of every epilogue against the 8-wave
    // kernel over 2,364 tiles x 9-10 tiles per workgroup on random operands (tests/test_gemm_nt_w128_gpu.py, tools/nt_probe): a stale stage would
    // show as a wrong tile.  dbg bit 1 (VITK_NTW_RELAX=n in the experiments build) makes every wait strict.
    // [measured, tools/nt_probe, strict vs exact-count waits: plain 16-bit stores (QKV) 155 vs 151 us; every epilogue that also READS rows
    //  or stores two tensors is level or better strict (FF1 shape 216 vs 231, dFF1 271 vs 275, out-projection 69.9 vs 76.6)]
    constexpr bool RELAX_OK = (EPI == VITK_EPI_NONE || (EPI == VITK_EPI_BIAS && !AW));      // (AW = 1: the bias loads of a tile sit behind the stores)
    bool relax = false;
#define V_WAITR do { \
        if (relax) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(VM_RELAX) : "memory"); \
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); \
    } while (0)

    for (int idx = l0; idx < count; idx += L) {
        int m0, n0, mt;
        decode(idx, m0, n0, mt);
        if constexpr (AW && q_has_bias<EPI>()) {
            if (p.bias) {       // the lane's 2 x 4 bias values of this tile (older than every piece issued from here on: the counted waits cover them)
                const __bf16* bp = p.bias + n0 + wn * 128 + 4 * (lane & 15);
                v_gload_bf16x4(bq[0], bp);
                v_gload_bf16x4(bq[1], bp + 64);
            }
        }
        // K-steps in pairs (the two fragment sets swap roles); the first K-step writes the accumulators with C = 0
        V_STEP(true, xa, wa, xb, wb, V_WAITR, 0);
        V_STEP(false, xb, wb, xa, wa, V_WAITR, 1);
        if constexpr (AW) {
            for (int kt = 2; kt + 6 < p.nt; kt += 2) {
                V_STEP(false, xa, wa, xb, wb, V_WAIT16, 0);
                V_STEP(false, xb, wb, xa, wa, V_WAIT16, 1);
            }
            V_STEP(false, xa, wa, xb, wb, V_WAIT16, 0);     // K-step nt - 6: completes the tile's last pair
            setup_a(idx + L);                               // the next three pairs are the next tile's first three ...
            V_STEP(false, xb, wb, xa, wa, V_WAIT16, 1);
            setup_w(idx + L);                               // ... and the W pieces of the last four K-steps its first four
            V_STEP(false, xa, wa, xb, wb, V_WAIT16, 0);
            V_STEP(false, xb, wb, xa, wa, V_WAIT16, 1);
            V_STEP(false, xa, wa, xb, wb, V_WAIT16, 0);
            V_STEP(false, xb, wb, xa, wa, V_WAIT16, 1);       // reads the NEXT tile's first fragments
        } else {
            for (int kt = 2; kt + 4 < p.nt; kt += 2) {
                V_STEP(false, xa, wa, xb, wb, V_WAIT16, 0);
                V_STEP(false, xb, wb, xa, wa, V_WAIT16, 1);
            }
            next_src(idx + L);       // the pieces of the last four K-steps are the next tile's first four
            V_STEP(false, xa, wa, xb, wb, V_WAIT16, 0);
            V_STEP(false, xb, wb, xa, wa, V_WAIT16, 1);
            V_STEP(false, xa, wa, xb, wb, V_WAIT16, 0);
            V_STEP(false, xb, wb, xa, wa, V_WAIT16, 1);       // reads the NEXT tile's first fragments
        }
        // the asm MFMAs' results are complete before the compiler's reads of them (it does not see the MFMAs' latency); nothing is
        // scheduled across the pin
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        V_PIN();
        epilogue(m0, n0, mt);
        relax = (RELAX_OK || (p.dbg & 16)) && !(p.dbg & 3);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the surplus DMA pieces must not outlive the workgroup's LDS allocation
#undef V_WAITR
#undef V_WAIT16
#undef V_STEP
#undef V_GROUP
#undef V_MFMA32
}

template <typename Kern>
int v_set_max_lds(Kern kernel, int bytes) {
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

}  // namespace

// Shapes the four-wave kernel takes: K-steps in pairs with a peeled first pair and four peeled last steps (K % 64 == 0, K >= 256), whole
// 256-column tiles (N % 256 == 0), 32-bit descriptor offsets.  It computes FULL 256-row tiles only, rows [0, 256 tiles_m): the caller
// (gemm_bf16.hip) gives the remaining rows -- a partial last m-tile, and the rows of a mostly idle last round, which the 8-wave kernel
// cuts into 128-row tiles -- to the 8-wave kernel in a second launch.
bool gemm_ntw_serves(int64_t M, int64_t N, int64_t K) {
    return !((K % 64) || K < 256 || (N % 256) || M < 256 || N > 16128 || (K / 32) * (long long)V_TILE >= (1LL << 31));
}

// how many of the floor(M / 256) full m-tiles go to the four-wave launch.  Rounds = tiles / resident workgroups; a last round that is
// mostly idle is cheaper as 128-row tiles of the 8-wave kernel -- if a tile is long enough to pay for the second launch.  Cost model in
// units of one 256-row tile of the four-wave kernel (~0.72 us per K-step + 6 us): a 128-row tile of the 8-wave kernel ~0.6, the second
// launch ~14 us [measured, tools/nt_probe: FF1 (K = 768, 9.23 rounds) 216 us in one launch vs 230 split; FF2 (K = 3072, 2.31 rounds) 226 vs
// 220; the out-projection (K = 768, 2.31 rounds) 70.0 vs 72.1].
int gemm_ntw_split(int64_t M, int64_t N, int64_t K, int grid) {
    const long long tiles_n = N / 256;
    // the rows left over go to the 8-wave persistent kernel, which takes M >= 1024: none, or at least 1024
    const long long tm_cap = (M % 256 == 0) ? M / 256 : (M - 1024) / 256;
    if (tm_cap <= 0) return 0;
    double second_launch = 14.0 / (0.72 * (double)(K / 32) + 6.0);
    if (const char* e = vitk_exp("VITK_NTW_SPLIT")) second_launch = e[0] == 'r' ? 0.15 : (e[0] == 'n' ? 100.0 : second_launch);      // A/B: rounds / none
    auto legal = [&](long long tm) { const long long rest = M - 256 * tm; return tm > 0 && tm <= tm_cap && (rest == 0 || rest >= 1024); };
    auto cost = [&](long long tm) -> double {
        const long long rest = M - 256 * tm;
        double c = (double)((tm * tiles_n + grid - 1) / grid);
        if (rest > 0) c += second_launch + 0.6 * (double)((((rest + 127) / 128) * tiles_n + grid - 1) / grid);
        return c;
    };
    long long best_tm = 0;
    double best = 1e30;
    auto consider = [&](long long tm) { if (legal(tm)) { const double c = cost(tm); if (c < best) { best = c; best_tm = tm; } } };
    consider(tm_cap);
    if (M % 256 == 0 && tm_cap > 4) consider(tm_cap - 4);
    const long long k = (tm_cap * tiles_n) / grid;          // whole rounds
    if (k >= 1) { long long tm = (k * grid) / tiles_n; consider(tm); while (tm > 0 && !legal(tm)) --tm; consider(tm); }
    if (vitk_exp("VITK_NTW_TM")) { const long long v = atoll(vitk_exp("VITK_NTW_TM")); if (v == 0 || legal(v)) best_tm = v; }    // experiments
    return (int)best_tm;
}

int gemm_ntw_launch(int tiles_m, int grid, const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                    int64_t N, int64_t K, int epilogue, const void* bias, const void* resid, void* aux, float* csum, int abl, int dbg, void* stream) {
    NtwArgs a;
    a.A = (const char*)A; a.lda = lda; a.W = (const char*)W; a.ldw = ldw; a.C = C; a.ldc = ldc;
    a.M = 256 * tiles_m; a.N = (int)N; a.K = (int)K;
    a.bias = (const __bf16*)bias; a.resid = resid; a.aux = (__bf16*)aux; a.csum = csum;
    a.tiles_n = (int)(N / 256); a.tiles_m = tiles_m; a.n_tiles = a.tiles_m * a.tiles_n; a.nt = (int)(K / 32);
    // grouped tile order (the 8-wave kernel's scheme): one group up to 8 n-tiles, beyond that groups of 4 (K <= 768) or balanced groups of <= 6; n fastest inside a group
    // [measured on the whole step, round 5, two boxes x 2-3 interleaved runs (profiles/r05g_group_sweep.log): groups of 4 n-tiles for the
    //  9- and 12-tile shapes (QKV, FF1, dFF1) 30.83 ms against 30.97 with the 8-wave kernel's balanced groups of <= 6; 3: 31.05; 9 / 12: level]
    //  With longer rows the balanced groups stay: ViT-L/16 (K = 1024) 52.44 ms with 6 against 52.58 with 4, ViT-H/14 (K = 1280) 664.5 ms with 5
    //  against 682.8 with 4 (profiles/r05g_group_sweep.log).]
    a.group_n = a.tiles_n;
    if (a.tiles_n > 8) a.group_n = K <= 768 ? 4 : (a.tiles_n + (a.tiles_n + 5) / 6 - 1) / ((a.tiles_n + 5) / 6);
    if (vitk_exp("VITK_GROUP_N")) { const int g = atoi(vitk_exp("VITK_GROUP_N")); a.group_n = g > 0 && g < a.tiles_n ? g : a.tiles_n; }
    a.dbg = dbg;
    if (const char* e = vitk_exp("VITK_NTW_RELAX")) a.dbg |= e[0] == 'a' ? 16 : (e[0] == 'n' ? 2 : 0);      // A/B: all / none
    const int lds_bytes = V_RING + a.tiles_n * 512 + 16;
    hipStream_t st = (hipStream_t)stream;
    if (tiles_m <= 0 || grid < 8) VITK_FAIL(VITK_E_ARG, "gemm_nt_bf16 (w128): nothing to do");
    // the activation feed: 128-byte rows (round 6) unless VITK_NTW_A128=0 asks for round 5's 64-byte pieces (the same-box A/B of the two feeds)
    const char* aw_e = vitk_switch("VITK_NTW_A128");
    const bool a128 = !(aw_e && aw_e[0] == '0');
#define NTW_LAUNCH1(E, AB) do { \
        if (a128) { \
            static const int rc1__ = v_set_max_lds(gemm_ntw_kernel<E, AB, 1>, X_LDS); \
            if (rc1__ != 0) VITK_FAIL(rc1__, "gemm_nt_bf16 (w128): cannot enable %d B of LDS", X_LDS); \
            hipLaunchKernelGGL((gemm_ntw_kernel<E, AB, 1>), dim3((unsigned)grid), dim3(256), X_LDS, st, a); \
        } else { \
            static const int rc__ = v_set_max_lds(gemm_ntw_kernel<E, AB, 0>, V_LDS_MAX); \
            if (rc__ != 0) VITK_FAIL(rc__, "gemm_nt_bf16 (w128): cannot enable %d B of LDS", V_LDS_MAX); \
            hipLaunchKernelGGL((gemm_ntw_kernel<E, AB, 0>), dim3((unsigned)grid), dim3(256), lds_bytes, st, a); \
        } \
    } while (0)
#ifdef NTW_PROBE
#define NTW_LAUNCH_ALL(E) do { \
        switch (abl) { \
            case 0: NTW_LAUNCH1(E, 0); break; case 1: NTW_LAUNCH1(E, 1); break; case 2: NTW_LAUNCH1(E, 2); break; case 3: NTW_LAUNCH1(E, 3); break; \
            case 4: NTW_LAUNCH1(E, 4); break; case 5: NTW_LAUNCH1(E, 5); break; case 6: NTW_LAUNCH1(E, 6); break; case 7: NTW_LAUNCH1(E, 7); break; \
            case 8: NTW_LAUNCH1(E, 8); break; case 10: NTW_LAUNCH1(E, 10); break; case 11: NTW_LAUNCH1(E, 11); break; \
            default: VITK_FAIL(VITK_E_ARG, "gemm_nt_bf16 (w128): bad ablation %d", abl); \
        } } while (0)
#define NTW_LAUNCH(E) do { if (abl == 0) NTW_LAUNCH1(E, 0); else if (abl == 7) NTW_LAUNCH1(E, 7); else VITK_FAIL(VITK_E_ARG, "gemm_nt_bf16 (w128): ablation %d exists for EPI_NONE only", abl); } while (0)
#else
#define NTW_LAUNCH(E) do { if (abl) VITK_FAIL(VITK_E_ARG, "gemm_nt_bf16 (w128): ablations exist in tools/nt_probe.hip only"); NTW_LAUNCH1(E, 0); } while (0)
#define NTW_LAUNCH_ALL(E) NTW_LAUNCH(E)
#endif
    switch (epilogue) {
        case VITK_EPI_NONE: NTW_LAUNCH_ALL(VITK_EPI_NONE); break;
#ifndef NTW_PROBE_LEAN
        case VITK_EPI_BIAS: NTW_LAUNCH(VITK_EPI_BIAS); break;
        case VITK_EPI_BIAS_GELU: NTW_LAUNCH(VITK_EPI_BIAS_GELU); break;
        case VITK_EPI_RESID: NTW_LAUNCH(VITK_EPI_RESID); break;
        case VITK_EPI_GELU_BWD: NTW_LAUNCH(VITK_EPI_GELU_BWD); break;
        case VITK_EPI_BIAS_GELU_DG: NTW_LAUNCH(VITK_EPI_BIAS_GELU_DG); break;
        case VITK_EPI_MUL_AUX: NTW_LAUNCH(VITK_EPI_MUL_AUX); break;
        case VITK_EPI_BIAS_GELU_DG8: NTW_LAUNCH(VITK_EPI_BIAS_GELU_DG8); break;
        case VITK_EPI_MUL_AUX8: NTW_LAUNCH(VITK_EPI_MUL_AUX8); break;
        case VITK_EPI_RESID16: NTW_LAUNCH(VITK_EPI_RESID16); break;
#endif
        default: VITK_FAIL(VITK_E_ARG, "gemm_nt_bf16 (w128): bad epilogue %d", epilogue);
    }
#undef NTW_LAUNCH_ALL
#undef NTW_LAUNCH
#undef NTW_LAUNCH1
    VITK_CHECK_LAUNCH("gemm_nt_bf16 (w128)");
    return 0;
}
