// tools/gemm_nt_x2.hip -- EXPERIMENT of round 6 (correct, bit-identical, NOT faster: profiles/r06u_nt_probe_two_groups_one_wg.log; not part of libvitk) -- persistent NT GEMM:  C[M,N] = A[M,K] . W[N,K]^T + fused epilogue, for the EPILOGUE-HEAVY shapes
// (FF1: bias + GELU + gelu' codes, vit.py:20-21; dFF1: x gelu' + bias-gradient column sums, the autograd of it).
//
// In the four-wave kernel (gemm_nt_w128.hip) a tile's epilogue runs after its main loop: at the FF1 shape 113 us of VALU-bound epilogue behind
// 162 us of main loop, with the matrix cores idle in the one and the VALU / store path idle in the other.  Everything tried to overlap them had
// failed for a measured reason: stores issued from inside the next main loop queue behind the wave's LDS-DMA pieces in the in-order vmcnt
// (round 5); two independent workgroups per CU do not stay out of phase and need 1.5x the feed (round 6, tools/gemm_nt_duo.hip).  Here the two
// halves of a CU's work are TWO WAVE GROUPS OF ONE WORKGROUP that swap roles under the workgroup's own barrier:
//   * 8 waves = 2 groups x 4; a group owns a 256 x 128 "virtual tile" (wave tile 128 x 64: 32 accumulator tiles in 128 AGPRs, two fragment
//     sets in 96 VGPRs: a wave fits 256 registers, two waves per SIMD);
//   * phase p: group p & 1 runs the MAIN LOOP of virtual tile p (MFMAs + fragment reads + ALL of the workgroup's LDS-DMA), the other group
//     the EPILOGUE of virtual tile p - 1 (VALU + global loads / stores), cut into one piece per K-step; every K-step ends in the ONE
//     s_barrier both groups execute, so the alternation cannot drift, and on every SIMD an MFMA wave sits beside a VALU wave;
//   * vmcnt is per wave: the DMA pieces are counted in the main-role waves, the stores in the epilogue-role waves -- no store acknowledge
//     ever sits in front of a DMA wait (at a role switch the old producer waits for its last two K-steps of pieces by exact counts before
//     its first store);
//   * one LDS ring for the workgroup (only one group is in a main loop at a time): three 32 KiB slots of 128-byte activation rows (a K-step
//     pair each: gemm_nt_w128.hip, AW = 1) + four 8 KiB stages of the K-blocked W image (one 128-column half of a 256-row block); the
//     stream is continuous across virtual tiles: the last K-steps of a phase fetch the OTHER group's first;
//   * accumulators, fragments, MFMA order and epilogue arithmetic are the four-wave kernel's: bit-identical results.
// Costs 1.5x the operand bytes per flop through the feed (256 x 128 tiles) -- which the 128-byte activation feed has room for at K <= 1024
// (FF1 shape: LDS-DMA alone 83 us x 1.5 against 108 us of MFMA) and not at K = 3072; served: the FF1 / dFF1 pair.
#include "../vit_pytorch_amd/csrc/common.h"
#include "../vit_pytorch_amd/csrc/gemm_nt_plan.h"
#include "../vit_pytorch_amd/csrc/gemm_nt_epi.h"
#include <stdlib.h>
#include <type_traits>

#ifdef VITK_HALF_IS_F16
#define NTX_MFMA_ASM "v_mfma_f32_16x16x32_f16"
#else
#define NTX_MFMA_ASM "v_mfma_f32_16x16x32_bf16"
#endif

namespace {

constexpr int X2_ASLOT = 256 * 128;                 // activation rows of a K-step PAIR: 256 rows of 128 bytes
constexpr int X2_WSTAGE = 128 * 64;                 // weight rows of a K-step: one 128-column half of a K-blocked block
constexpr int X2_WBLOCK = 256 * 64;                 // a K-blocked W block (vitk_pack_w_nt)
constexpr int X2_WBASE = 3 * X2_ASLOT;              // 96 KiB
constexpr int X2_LDS = X2_WBASE + 4 * X2_WSTAGE;    // 128 KiB
constexpr int X2_EPI_SLOT0 = 2;                     // first K-step slot of a phase that carries an epilogue piece
constexpr int X2_EPI_PARTS = 16;                    // 8 fragment rows x 2 row pairs
constexpr int X2_MIN_NT = X2_EPI_SLOT0 + X2_EPI_PARTS + 4;      // + store-free slots before the role switch

#define X2_PIN() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ int x2_swz(int x) { return (0x1320 >> (4 * (x & 3))) & 3; }   // permutation [0,2,3,1] (the W image's)

__device__ __forceinline__ void x2_grouped_tile(int t, int tiles_m, int tiles_n, int gn, int& tm, int& tn) {
    const int per_group = gn * tiles_m;
    const int g = t / per_group;
    const int r = t - g * per_group;
    const int rem = tiles_n - g * gn;
    const int w = rem < gn ? rem : gn;
    tm = r / w;
    tn = g * gn + (r - tm * w);
}

struct NtxArgs {
    const char* A; long long lda;      // element strides; operands are 2-byte elements
    const char* W; long long ldw;      // ldw == 0: K-blocked (vitk_pack_w_nt)
    void* C; long long ldc;
    int M, N, K;
    const __bf16* bias; const void* resid; __bf16* aux; float* csum;
    int tiles_n, group_n, tiles_m, n_tiles, nt;      // 128-column virtual tiles; FULL interior tiles only: rows [0, 256 tiles_m), N % 256 == 0
    int dbg;            // experiments: bit 0 = no epilogue pieces (the barriers stay): main loops alone
};

template <int OFF> __device__ __forceinline__ bf16x8 x2_rd(unsigned lds_addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(OFF) : "memory");
    return v;
}
template <bool Z> __device__ __forceinline__ void x2_mfma(f32x4& c, const bf16x8& a, const bf16x8& b) {
    if constexpr (Z) asm volatile("" NTX_MFMA_ASM " %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
    else asm volatile("" NTX_MFMA_ASM " %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void x2_gload_bf16x4(bf16x4& d, const __bf16* p) { asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }

// ABL (experiments, tools/nt_probe.hip): bit 0 no LDS-DMA, bit 1 no fragment reads, bit 2 no MFMA
template <int EPI, int ABL>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_ntx_kernel(const NtxArgs p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, w4 = wave & 3;
    const int wm = w4 >> 1, wn = w4 & 1;

    // ---- this workgroup's virtual tiles: XCD x owns a contiguous run of the list; its workgroups take every L-th tile of it ----
    const int xcd = blockIdx.x & 7, l0 = blockIdx.x >> 3, L = gridDim.x >> 3;
    const int ms = (int)(((long long)xcd * p.n_tiles) >> 3), count = (int)(((long long)(xcd + 1) * p.n_tiles) >> 3) - ms;
    if (l0 >= count) return;
    const int V = (count - l0 + L - 1) / L;             // virtual tiles of this workgroup: phases 0 .. V
    auto decode = [&](int idx, int& m0, int& n0, int& mt) {
        int tn;
        x2_grouped_tile(ms + idx, p.tiles_m, p.tiles_n, p.group_n, mt, tn);
        m0 = mt * 256; n0 = tn * 128;
    };

    // ---- producer state: kept by EVERY wave (scalar), used by the waves of the group in the main role ----
    // activation slot (a K-step pair): an instruction fills 8 LDS rows of 128 bytes; wave w4 of the producing group owns pieces 8 w4 .. 8 w4 + 7,
    // piece i = rows 64 w4 + 8 i + (lane >> 3); LDS row R, position s holds logical chunk s ^ ((R >> 1) & 7) = s ^ (4 (i & 1) + (lane >> 4)).
    // W stage: wave w4 owns pieces 2 w4, 2 w4 + 1; LDS row R = 64 q + 16 fn + c holds W row 64 q + 4 c + fn of the 128-column tile.
    const int srow = lane >> 2, spos = lane & 3;
    const int schunk = spos ^ x2_swz(lane >> 4);
    int avo[2], wvo[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        avo[j] = (int)(((long long)(64 * w4 + (lane >> 3)) * p.lda + (((lane & 7) ^ (4 * j + (lane >> 4))) * 8)) * 2);      // j = piece parity
        wvo[j] = p.ldw == 0 ? (2 * w4 + j) * 1024 + lane * 16
                            : (int)(((long long)(64 * (w4 >> 1) + 4 * srow + 2 * (w4 & 1) + j) * p.ldw + schunk * 8) * 2);
    }
    const int a_rows8 = (int)(p.lda * 16);
    const int w_kstride = p.ldw == 0 ? X2_WBLOCK : 64;
    __amdgpu_buffer_rsrc_t a_rs, w_rs;
    int a_so = 0, w_so = 0;
    auto setup_a = [&](int idx) __attribute__((always_inline)) {
        if (idx < count) {
            int m0, n0, mt;
            decode(idx, m0, n0, mt);
            const long long abytes = (255LL * p.lda + p.K) * 2;
            a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (long long)m0 * p.lda * 2), 0, (int)abytes, 0x00020000);
        } else a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0, 0x00020000);       // past the list: range 0 (pieces count, fetch nothing)
        a_so = 0;
    };
    auto setup_w = [&](int idx) __attribute__((always_inline)) {
        w_so = 0;
        if (idx < count) {
            int m0, n0, mt;
            decode(idx, m0, n0, mt);
            if (p.ldw == 0) {
                w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (long long)(n0 >> 8) * p.nt * X2_WBLOCK), 0, p.nt * X2_WBLOCK, 0x00020000);
                w_so = ((n0 >> 7) & 1) * X2_WSTAGE;
            } else {
                const long long wbytes = (127LL * p.ldw + p.K) * 2;
                w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (long long)n0 * p.ldw * 2), 0, (int)wbytes, 0x00020000);
            }
        } else w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0, 0x00020000);
    };
    auto dma_a = [&](int slot, int i) __attribute__((always_inline)) {
        char* dst = lds + slot * X2_ASLOT + (w4 * 8 + i) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (void __attribute__((address_space(3)))*)dst, 16, avo[i & 1], a_so + i * a_rows8, 0, 0);
    };
    auto dma_w = [&](int stg_, int j) __attribute__((always_inline)) {
        char* dst = lds + X2_WBASE + stg_ * X2_WSTAGE + (w4 * 2 + j) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (void __attribute__((address_space(3)))*)dst, 16, wvo[j], w_so, 0, 0);
    };

    // ---- consumer addresses ----
    const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)lds);
    unsigned a_rd, a_rd1, w_rd;                          // + f * 2048 / + fn * 1024, + slot / stage
    {
        const int fi = lane & 15, fg = lane >> 4;
        const int fpos = fg ^ x2_swz(fi >> 2);
        const int xpos = fg ^ ((fi >> 1) & 7);
        a_rd = lds_base + (wm * 128 + fi) * 128 + xpos * 16;
        a_rd1 = lds_base + (wm * 128 + fi) * 128 + (xpos ^ 4) * 16;
        w_rd = lds_base + X2_WBASE + (wn * 64 + fi) * 64 + fpos * 16;
    }

    int stg = 0;                // K-step counter mod 4: the W stage whose fragments are in registers
    int sp = 0;                 // K-step pair counter mod 3: the activation slot of the pair the MFMAs are in
    bf16x8 xa[8], wa[4], xb[8], wb[4];
    f32x4 acc[4][8];            // acc[fn][f][j]: row 16 f + 4 fg + j, column 4 fi + fn of the wave tile

    // after K-step t of a phase, in EVERY wave (both roles): the barrier, then the stream position of K-step t + 1.  The producer changes virtual
    // tile at fixed K-steps: the activation stream (three pairs ahead) before K-step nt - 5, the W stream (four K-steps ahead) before nt - 4.
    auto post = [&](int t, int h, int next_idx) __attribute__((always_inline)) {
        X2_PIN();
        __builtin_amdgcn_s_barrier();
        X2_PIN();
        w_so += w_kstride;
        if (!h) a_so += 128;
        else sp = sp == 2 ? 0 : sp + 1;
        stg = (stg + 1) & 3;
        if (t == p.nt - 6) setup_a(next_idx);
        if (t == p.nt - 5) setup_w(next_idx);
    };

    // ---- main role: one K-step.  H = 0: reads the pair's second K-step (slot sp, positions ^ 4), issues the second half of pair P + 2 into slot
    //      sp + 2; H = 1: reads the first K-step of pair P + 1 (slot sp + 1), issues the first half of pair P + 3 into slot sp.  W pieces of
    //      K-step t + 4 into stage stg.  Groups: activation fragment G x the four W fragments. ----
#define X2_GROUP(G, Z, XC, WC, XN, WN) do { \
        if constexpr (!(ABL & 4)) x2_mfma<Z>(acc[0][(G)], XC[(G)], WC[0]); \
        if constexpr (!(ABL & 2)) { if constexpr ((G) < 4) XN[2 * (G)] = x2_rd<(2 * (G)) * 2048>(rdA); else WN[(G) - 4] = x2_rd<((G) - 4) * 1024>(rdW); } \
        if constexpr (!(ABL & 4)) x2_mfma<Z>(acc[1][(G)], XC[(G)], WC[1]); \
        if constexpr (!(ABL & 2)) { if constexpr ((G) < 4) XN[2 * (G) + 1] = x2_rd<(2 * (G) + 1) * 2048>(rdA); } \
        if constexpr (!(ABL & 4)) x2_mfma<Z>(acc[2][(G)], XC[(G)], WC[2]); \
        if constexpr (!(ABL & 1) && ((G) & 3) != 0) { X2_PIN(); \
            if constexpr ((G) == 1 || (G) == 2 || (G) == 3) dma_a(dslot_, di0_ + (G) - 1); \
            else if constexpr ((G) == 5) dma_a(dslot_, di0_ + 3); \
            else dma_w(stg, (G) - 6); \
            X2_PIN(); } \
        if constexpr (!(ABL & 4)) x2_mfma<Z>(acc[3][(G)], XC[(G)], WC[3]); \
    } while (0)
#define X2_STEP(Z, XC, WC, XN, WN, H) do { \
        const int sp1_ = sp == 2 ? 0 : sp + 1; \
        const int dslot_ = (H) ? sp : (sp == 0 ? 2 : sp - 1); \
        constexpr int di0_ = (H) ? 0 : 4; \
        const unsigned rdA = (H) ? a_rd + sp1_ * X2_ASLOT : a_rd1 + sp * X2_ASLOT; \
        const unsigned rdW = w_rd + ((stg + 1) & 3) * X2_WSTAGE; \
        __builtin_amdgcn_s_setprio(1); \
        X2_GROUP(0, Z, XC, WC, XN, WN); X2_GROUP(1, Z, XC, WC, XN, WN); X2_GROUP(2, Z, XC, WC, XN, WN); X2_GROUP(3, Z, XC, WC, XN, WN); \
        X2_GROUP(4, Z, XC, WC, XN, WN); X2_GROUP(5, Z, XC, WC, XN, WN); X2_GROUP(6, Z, XC, WC, XN, WN); X2_GROUP(7, Z, XC, WC, XN, WN); \
        __builtin_amdgcn_s_setprio(0); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        if constexpr (!(ABL & 1)) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");       /* own pieces of K-step t + 2 landed (t + 3, t + 4 fly) */ \
    } while (0)

    // ---- epilogue role: the pieces of one virtual tile.  piece k = fragment row f = k >> 1 (4 output rows per lane), row pair pr = k & 1 ----
    // (the other NT kernels' arithmetic per 64-column block; operand rows by uncounted asm loads DP rows ahead, waited for by exact counts over
    //  the epilogue's OWN memory operations: this wave's LDS-DMA pieces of its last main loop are older than all of them and were waited for in
    //  the first two slots)
    constexpr bool AUX_IN = q_aux_in<EPI>();
    constexpr bool AUX8 = (EPI == VITK_EPI_MUL_AUX8);
    constexpr bool HAS_BIAS = q_has_bias<EPI>();
    constexpr int NR = 8;
    constexpr int DP = AUX8 ? 4 : 3;
    using HPre = std::conditional_t<AUX8, q_u32x2, bf16x8>;
    HPre hpre[AUX_IN ? DP : 1][2];
    bf16x4 bq = bf16x4{0, 0, 0, 0};
    float cs[8];
    long long e_obase = 0;      // element (row mrow0 + odd, column ncolw + 8 (fi >> 1)) of the tile under the epilogue
    int e_odd = 0, e_mt = 0, e_ncol = 0, e_fg = 0, e_col8 = 0;
    auto fetch_pre = [&](int f, HPre (&dst)[2]) __attribute__((always_inline)) {
        long long oa = e_obase + (long long)(f * 16) * p.ldc;
        asm volatile("" : "+v"(oa));
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            if constexpr (AUX8) q_gload_u32x2(dst[pr], reinterpret_cast<const unsigned char*>(p.aux) + oa + (long long)(2 * pr) * p.ldc);
            else q_gload_bf16x8(dst[pr], p.aux + oa + (long long)(2 * pr) * p.ldc);
        }
    };
    // slot 0 of an epilogue phase: the tile's coordinates, its first operand rows and bias values (E0 memory operations)
    constexpr int E0 = (HAS_BIAS ? 1 : 0) + (AUX_IN ? 2 * DP : 0);
    auto epi_begin = [&](int idx) __attribute__((always_inline)) {
        int m0, n0, mt;
        decode(idx, m0, n0, mt);
        int elane;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(elane));
        const int fi = elane & 15, fg = elane >> 4;
        e_odd = fi & 1; e_mt = mt; e_fg = fg;
        e_ncol = n0 + wn * 64;
        e_col8 = e_ncol + 8 * (fi >> 1);
        e_obase = (long long)(m0 + wm * 128 + 4 * fg + e_odd) * p.ldc + e_ncol + 8 * (fi >> 1);
        if constexpr (HAS_BIAS) {       // (one load either way: the exact counts below include it; a null bias reads any valid address and is zeroed at its use)
            x2_gload_bf16x4(bq, p.bias ? p.bias + e_ncol + 4 * fi : reinterpret_cast<const __bf16*>(p.A));
        }
        if constexpr (AUX_IN) {
#pragma unroll
            for (int i = 0; i < DP; ++i) fetch_pre(i, hpre[i]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[e] = 0.f;
    };
    auto epi_part = [&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        constexpr int f = k >> 1, pr = k & 1;
        if (p.dbg & 1) {
            if constexpr (k == X2_EPI_PARTS - 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) asm volatile("" :: "a"(acc[i][j]));
            }
            return;
        }
        __bf16* Cb = reinterpret_cast<__bf16*>(p.C);
        if constexpr (pr == 0) {
            X2_PIN();
            asm volatile("" : "+a"(acc[0][f]), "+a"(acc[1][f]), "+a"(acc[2][f]), "+a"(acc[3][f]));      // (the AGPR -> VGPR copies are formed here)
            if constexpr (AUX_IN) q_wait_regs2<q_epi_younger(f, NR, DP, 2)>(hpre[f % DP][0], hpre[f % DP][1]);
            else if constexpr (HAS_BIAS && k == 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(bq) :: "memory");      // the bias load of slot 0 (nothing else is in flight)
        }
        f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (HAS_BIAS) { if (p.bias) b4 = f32x4{(float)bq[0], (float)bq[1], (float)bq[2], (float)bq[3]}; }
        long long o0 = e_obase + (long long)(f * 16) * p.ldc;
        asm volatile("" : "+v"(o0));
        const int odd = e_odd;
        constexpr int j0 = 2 * pr;
        const unsigned a0 = q_pack2(acc[0][f][j0] + b4[0], acc[1][f][j0] + b4[1]);
        const unsigned a1 = q_pack2(acc[2][f][j0] + b4[2], acc[3][f][j0] + b4[3]);
        const unsigned c0 = q_pack2(acc[0][f][j0 + 1] + b4[0], acc[1][f][j0 + 1] + b4[1]);
        const unsigned c1 = q_pack2(acc[2][f][j0 + 1] + b4[2], acc[3][f][j0 + 1] + b4[3]);
        const unsigned r0 = q_dpp_xor1(odd ? a0 : c0), r1 = q_dpp_xor1(odd ? a1 : c1);
        const unsigned k0 = odd ? c0 : a0, k1 = odd ? c1 : a1;
        const q_u32x4 w4v = odd ? q_u32x4{r0, r1, k0, k1} : q_u32x4{k0, k1, r0, r1};
        const bf16x8 v = __builtin_bit_cast(bf16x8, w4v);
        const long long o = o0 + (long long)j0 * p.ldc;
        if constexpr (EPI == VITK_EPI_NONE || EPI == VITK_EPI_BIAS) {
            *reinterpret_cast<bf16x8*>(Cb + o) = v;
        } else if constexpr (EPI == VITK_EPI_BIAS_GELU_DG || EPI == VITK_EPI_BIAS_GELU_DG8) {
            q_f32x8 gl, dgl;
#ifdef NTX_PROBE
            if (p.dbg & 32) { gl = q_widen8(v); dgl = gl; } else      // experiments (timing only): bit 5 = no GELU arithmetic, bit 6 = no stores
#endif
            q_gelu_both8(q_widen8(v), gl, dgl);              // of the ROUNDED pre-activation
#ifdef NTX_PROBE
            if (p.dbg & 64) { const bf16x8 t0 = q_narrow8(gl); const q_u32x2 t1 = q_dg_encode8(dgl); asm volatile("" :: "v"(t0), "v"(t1)); } else {
#endif
            if constexpr (EPI == VITK_EPI_BIAS_GELU_DG8) *reinterpret_cast<q_u32x2*>(reinterpret_cast<unsigned char*>(p.aux) + o) = q_dg_encode8(dgl);
            else *reinterpret_cast<bf16x8*>(p.aux + o) = q_narrow8(dgl);
            *reinterpret_cast<bf16x8*>(Cb + o) = q_narrow8(gl);
#ifdef NTX_PROBE
            }
#endif
        } else if constexpr (AUX_IN) {
            q_f32x8 fac;
            if constexpr (AUX8) fac = q_dg_decode8(hpre[f % DP][pr]);
            else fac = q_widen8(hpre[f % DP][pr]);
            const q_f32x8 g = q_widen8(v) * fac;
            const bf16x8 g8 = q_narrow8(g);
            *reinterpret_cast<bf16x8*>(Cb + o) = g8;
#pragma unroll
            for (int e = 0; e < 8; ++e) cs[e] += (float)g8[e];     // of the ROUNDED values: what colsum(C) would read
        }
        if constexpr (AUX_IN && pr == 1) {
            asm volatile("" : "+v"(cs[0]), "+v"(cs[1]), "+v"(cs[2]), "+v"(cs[3]), "+v"(cs[4]), "+v"(cs[5]), "+v"(cs[6]), "+v"(cs[7]));
            if constexpr (f + DP < NR) fetch_pre(f + DP, hpre[f % DP]);
        }
        if constexpr (AUX_IN && k == X2_EPI_PARTS - 1) {
            if (p.csum) {       // bias gradient by-product: one partial row per (m-tile, wm), the 8 lanes that own the same 8 columns summed in registers
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float vv = cs[e];
                    vv += __builtin_bit_cast(float, q_dpp_xor1(__builtin_bit_cast(unsigned, vv)));
                    unsigned u = __builtin_bit_cast(unsigned, vv);
                    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
                    vv = __builtin_bit_cast(float, (unsigned)a[0]) + __builtin_bit_cast(float, (unsigned)a[1]);
                    u = __builtin_bit_cast(unsigned, vv);
                    auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                    cs[e] = __builtin_bit_cast(float, (unsigned)b[0]) + __builtin_bit_cast(float, (unsigned)b[1]);
                }
                if (e_fg == 0 && !odd) {
                    float* cp = p.csum + (long long)(2 * e_mt + wm) * p.N + e_col8;
                    *reinterpret_cast<f32x4*>(cp) = f32x4{cs[0], cs[1], cs[2], cs[3]};
                    *reinterpret_cast<f32x4*>(cp + 4) = f32x4{cs[4], cs[5], cs[6], cs[7]};
                }
            }
        }
    };

    // ---- prologue (group 0 produces): pairs 0, 1 and the first half of pair 2, W K-steps 0..3; in issue order pair 0, W 0, W 1 | pair 1, W 2, W 3 | half ----
    setup_a(l0); setup_w(l0);
    {
        const bool prod = grp == 0;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            if constexpr (!(ABL & 1)) { if (prod) {
#pragma unroll
                for (int i = 0; i < 8; ++i) dma_a(pr, i);
            } }
            a_so += 128;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                if constexpr (!(ABL & 1)) { if (prod) { dma_w(2 * pr + kk, 0); dma_w(2 * pr + kk, 1); } }
                w_so += w_kstride;
            }
        }
        if constexpr (!(ABL & 1)) { if (prod) {
#pragma unroll
            for (int i = 0; i < 4; ++i) dma_a(2, i);
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");       // pair 0, W 0, W 1 landed
        } }
        X2_PIN();
        __builtin_amdgcn_s_barrier();
        X2_PIN();
        if (prod) {
            const unsigned rdA = a_rd, rdW = w_rd;
            xa[0] = x2_rd<0 * 2048>(rdA); xa[1] = x2_rd<1 * 2048>(rdA); xa[2] = x2_rd<2 * 2048>(rdA); xa[3] = x2_rd<3 * 2048>(rdA);
            xa[4] = x2_rd<4 * 2048>(rdA); xa[5] = x2_rd<5 * 2048>(rdA); xa[6] = x2_rd<6 * 2048>(rdA); xa[7] = x2_rd<7 * 2048>(rdA);
            wa[0] = x2_rd<0 * 1024>(rdW); wa[1] = x2_rd<1 * 1024>(rdW); wa[2] = x2_rd<2 * 1024>(rdW); wa[3] = x2_rd<3 * 1024>(rdW);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        X2_PIN();
        __builtin_amdgcn_s_barrier();           // slot 0 / stage 0 have been read: K-step 0 may refill them
        X2_PIN();
    }
    if constexpr (ABL & 2) {
#pragma unroll
        for (int f = 0; f < 8; ++f) xb[f] = xa[f];
#pragma unroll
        for (int f = 0; f < 4; ++f) wb[f] = wa[f];
    }

    // ---- phases ----
    for (int ph = 0; ph <= V; ++ph) {
        const int idx = l0 + ph * L;                    // the virtual tile of this phase's main loop (ph < V)
        const int nxt = idx + L;                        // ... whose stream is followed by this one's
        const bool main_role = (ph & 1) == grp && ph < V;
        const bool epi_role = (ph & 1) != grp && ph >= 1;
        if (main_role) {
            X2_STEP(true, xa, wa, xb, wb, 0); post(0, 0, nxt);
            X2_STEP(false, xb, wb, xa, wa, 1); post(1, 1, nxt);
            for (int t = 2; t < p.nt; t += 2) {
                X2_STEP(false, xa, wa, xb, wb, 0); post(t, 0, nxt);
                X2_STEP(false, xb, wb, xa, wa, 1); post(t + 1, 1, nxt);
            }
            // the asm MFMAs' results are complete before the compiler's reads of them in the next phase (a whole barrier and more away)
        } else if (epi_role) {
            // slot 0: this wave's pieces of its last-but-one K-step landed (its last K-step's 6 and the E0 operations just issued may fly)
            epi_begin(idx - L);
            if constexpr (!(ABL & 1)) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(6 + E0) : "memory");
            post(0, 0, nxt);
            // slot 1: ... of its last K-step too
            if constexpr (!(ABL & 1)) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(E0) : "memory");
            post(1, 1, nxt);
            epi_part(std::integral_constant<int, 0>{}); post(2, 0, nxt);
            epi_part(std::integral_constant<int, 1>{}); post(3, 1, nxt);
            epi_part(std::integral_constant<int, 2>{}); post(4, 0, nxt);
            epi_part(std::integral_constant<int, 3>{}); post(5, 1, nxt);
            epi_part(std::integral_constant<int, 4>{}); post(6, 0, nxt);
            epi_part(std::integral_constant<int, 5>{}); post(7, 1, nxt);
            epi_part(std::integral_constant<int, 6>{}); post(8, 0, nxt);
            epi_part(std::integral_constant<int, 7>{}); post(9, 1, nxt);
            epi_part(std::integral_constant<int, 8>{}); post(10, 0, nxt);
            epi_part(std::integral_constant<int, 9>{}); post(11, 1, nxt);
            epi_part(std::integral_constant<int, 10>{}); post(12, 0, nxt);
            epi_part(std::integral_constant<int, 11>{}); post(13, 1, nxt);
            epi_part(std::integral_constant<int, 12>{}); post(14, 0, nxt);
            epi_part(std::integral_constant<int, 13>{}); post(15, 1, nxt);
            epi_part(std::integral_constant<int, 14>{}); post(16, 0, nxt);
            epi_part(std::integral_constant<int, 15>{}); post(17, 1, nxt);
            for (int t = 18; t < p.nt - 2; t += 2) { post(t, 0, nxt); post(t + 1, 1, nxt); }
            post(p.nt - 2, 0, nxt);
            // last slot: the first fragments of this group's NEXT main loop (the K-step after this phase), as a main-role K-step with H = 1 reads them
            {
                const int sp1_ = sp == 2 ? 0 : sp + 1;
                const unsigned rdA = a_rd + sp1_ * X2_ASLOT, rdW = w_rd + ((stg + 1) & 3) * X2_WSTAGE;
                xa[0] = x2_rd<0 * 2048>(rdA); xa[1] = x2_rd<1 * 2048>(rdA); xa[2] = x2_rd<2 * 2048>(rdA); xa[3] = x2_rd<3 * 2048>(rdA);
                xa[4] = x2_rd<4 * 2048>(rdA); xa[5] = x2_rd<5 * 2048>(rdA); xa[6] = x2_rd<6 * 2048>(rdA); xa[7] = x2_rd<7 * 2048>(rdA);
                wa[0] = x2_rd<0 * 1024>(rdW); wa[1] = x2_rd<1 * 1024>(rdW); wa[2] = x2_rd<2 * 1024>(rdW); wa[3] = x2_rd<3 * 1024>(rdW);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            post(p.nt - 1, 1, nxt);
        } else {
            // no role in this phase (group 1 in phase 0; the group without a tile in the last phase): the barriers and the stream position only --
            // and, for group 1 in phase 0, the first fragments of its main loop in the last slot
            for (int t = 0; t < p.nt - 2; t += 2) { post(t, 0, nxt); post(t + 1, 1, nxt); }
            post(p.nt - 2, 0, nxt);
            {
                const int sp1_ = sp == 2 ? 0 : sp + 1;
                const unsigned rdA = a_rd + sp1_ * X2_ASLOT, rdW = w_rd + ((stg + 1) & 3) * X2_WSTAGE;
                xa[0] = x2_rd<0 * 2048>(rdA); xa[1] = x2_rd<1 * 2048>(rdA); xa[2] = x2_rd<2 * 2048>(rdA); xa[3] = x2_rd<3 * 2048>(rdA);
                xa[4] = x2_rd<4 * 2048>(rdA); xa[5] = x2_rd<5 * 2048>(rdA); xa[6] = x2_rd<6 * 2048>(rdA); xa[7] = x2_rd<7 * 2048>(rdA);
                wa[0] = x2_rd<0 * 1024>(rdW); wa[1] = x2_rd<1 * 1024>(rdW); wa[2] = x2_rd<2 * 1024>(rdW); wa[3] = x2_rd<3 * 1024>(rdW);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            post(p.nt - 1, 1, nxt);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the surplus DMA pieces must not outlive the workgroup's LDS allocation
#undef X2_STEP
#undef X2_GROUP
}

template <typename Kern>
int x2_set_max_lds(Kern kernel, int bytes) {
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

}  // namespace

// Shapes this kernel takes: the epilogue of a virtual tile is spread over the K-steps of the next one (2 + 16 + 4 slots: K >= 704), K-steps in
// pairs (K % 64 == 0), whole 256-column blocks of W (N % 256 == 0), 32-bit descriptor offsets.  FULL 256-row tiles only, rows [0, 256 tiles_m).
bool gemm_ntx_serves(int64_t M, int64_t N, int64_t K) {
    return !((K % 64) || K / 32 < X2_MIN_NT || (N % 256) || M < 256 || N > 16128 || (K / 32) * (long long)X2_WBLOCK >= (1LL << 31));
}

int gemm_ntx_launch(int tiles_m, int grid, const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                    int64_t N, int64_t K, int epilogue, const void* bias, const void* resid, void* aux, float* csum, int abl, int dbg, void* stream) {
    NtxArgs a;
    a.A = (const char*)A; a.lda = lda; a.W = (const char*)W; a.ldw = ldw; a.C = C; a.ldc = ldc;
    a.M = 256 * tiles_m; a.N = (int)N; a.K = (int)K;
    a.bias = (const __bf16*)bias; a.resid = resid; a.aux = (__bf16*)aux; a.csum = csum;
    a.tiles_n = (int)(N / 128); a.tiles_m = tiles_m; a.n_tiles = a.tiles_m * a.tiles_n; a.nt = (int)(K / 32);
    // grouped tile order: the four-wave kernel's groups, in 128-column tiles
    const int tn256 = (int)(N / 256);
    int g256 = tn256;
    if (tn256 > 8) g256 = K <= 768 ? 4 : (tn256 + (tn256 + 5) / 6 - 1) / ((tn256 + 5) / 6);
    if (vitk_exp("VITK_GROUP_N")) { const int g = atoi(vitk_exp("VITK_GROUP_N")); g256 = g > 0 && g < tn256 ? g : tn256; }
    a.group_n = 2 * g256;
    a.dbg = dbg;
    hipStream_t st = (hipStream_t)stream;
    if (tiles_m <= 0 || grid < 8) VITK_FAIL(VITK_E_ARG, "gemm_nt_bf16 (x2): nothing to do");
    if (a.nt < X2_MIN_NT || (a.nt & 1)) VITK_FAIL(VITK_E_SHAPE, "gemm_nt_bf16 (x2): K = %lld not served", (long long)K);
#define NTX_LAUNCH1(E, AB) do { \
        static const int rc__ = x2_set_max_lds(gemm_ntx_kernel<E, AB>, X2_LDS); \
        if (rc__ != 0) VITK_FAIL(rc__, "gemm_nt_bf16 (x2): cannot enable %d B of LDS", X2_LDS); \
        hipLaunchKernelGGL((gemm_ntx_kernel<E, AB>), dim3((unsigned)grid), dim3(512), X2_LDS, st, a); \
    } while (0)
#ifdef NTX_PROBE
#define NTX_LAUNCH_ALL(E) do { \
        switch (abl) { \
            case 0: NTX_LAUNCH1(E, 0); break; case 1: NTX_LAUNCH1(E, 1); break; case 2: NTX_LAUNCH1(E, 2); break; case 3: NTX_LAUNCH1(E, 3); break; \
            case 4: NTX_LAUNCH1(E, 4); break; case 5: NTX_LAUNCH1(E, 5); break; case 6: NTX_LAUNCH1(E, 6); break; case 7: NTX_LAUNCH1(E, 7); break; \
            default: VITK_FAIL(VITK_E_ARG, "gemm_nt_bf16 (x2): bad ablation %d", abl); \
        } } while (0)
#define NTX_LAUNCH(E) do { if (abl == 0) NTX_LAUNCH1(E, 0); else if (abl == 7) NTX_LAUNCH1(E, 7); else VITK_FAIL(VITK_E_ARG, "gemm_nt_bf16 (x2): ablation %d exists for EPI_NONE only", abl); } while (0)
#else
#define NTX_LAUNCH(E) do { if (abl) VITK_FAIL(VITK_E_ARG, "gemm_nt_bf16 (x2): ablations exist in tools/nt_probe.hip only"); NTX_LAUNCH1(E, 0); } while (0)
#define NTX_LAUNCH_ALL(E) NTX_LAUNCH(E)
#endif
    switch (epilogue) {
#ifdef NTX_PROBE
        case VITK_EPI_NONE: NTX_LAUNCH_ALL(VITK_EPI_NONE); break;
#endif
        case VITK_EPI_BIAS_GELU_DG: NTX_LAUNCH(VITK_EPI_BIAS_GELU_DG); break;
        case VITK_EPI_MUL_AUX: NTX_LAUNCH(VITK_EPI_MUL_AUX); break;
        case VITK_EPI_BIAS_GELU_DG8: NTX_LAUNCH(VITK_EPI_BIAS_GELU_DG8); break;
        case VITK_EPI_MUL_AUX8: NTX_LAUNCH(VITK_EPI_MUL_AUX8); break;
        default: VITK_FAIL(VITK_E_ARG, "gemm_nt_bf16 (x2): epilogue %d not served", epilogue);
    }
#undef NTX_LAUNCH_ALL
#undef NTX_LAUNCH
#undef NTX_LAUNCH1
    VITK_CHECK_LAUNCH("gemm_nt_bf16 (x2)");
    return 0;
}
