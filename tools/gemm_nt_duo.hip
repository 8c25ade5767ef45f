// tools/gemm_nt_duo.hip -- EXPERIMENT of round 6 (measured, slower, not part of libvitk: profiles/r06a_ntd_probe_two_wg_per_cu.log) -- persistent NT GEMM:  C[M,N] = A[M,K] . W[N,K]^T (+ fused epilogue)
//
// nn.Linear forward (vit.py:20,23,44,47) and the dX GEMMs of its autograd -- the epilogue-heavy ones (FF1 + GELU, dFF1 x gelu', FF2 / out-projection
// + residual).  In the four-wave kernel (gemm_nt_w128.hip) the accumulators fill the register file (256 AGPRs), so a CU holds ONE workgroup and
// its matrix cores idle while that workgroup's epilogue stores a tile at the memory system's rate (~6 ms of a 30 ms step).  Here a CU holds
// TWO independent workgroups, each with its own barrier, its own vmcnt counters and its own LDS ring, started half a tile apart: while one is
// in its epilogue the other has the matrix cores to itself.
//   * tile 256 x 128 per workgroup, FOUR waves, one per SIMD and per workgroup (two waves per SIMD per CU), wave tile 128 x 64: the 32
//     accumulator tiles pinned in 128 AGPRs by asm MFMAs, two fragment sets (8 activation + 4 weight fragments each) in 96 VGPRs, the
//     whole wave within 256 registers;
//   * the same LDS image of a K-step as the other NT kernels (64-byte rows, chunk swizzle [0,2,3,1], W rows de-interleaved so that a lane's four
//     B fragments are four consecutive output columns; a 128-column tile is one contiguous half of the K-blocked 256-row W block), operands by
//     buffer-descriptor LDS-DMA, a ring of THREE 24 KiB stages per workgroup (2 x 72 KiB of the CU's 160), one barrier per K-step;
//   * the K-step stream continuous across tiles (the next tile's first three K-steps in flight during the epilogue);
//   * the epilogue arithmetic is the other kernels', per 64-column block: results are bit-identical to theirs.
// Which of a CU's two workgroups starts late: the hardware wave slot of its first wave (HW_REG_HW_ID.wave_id & 1; the second workgroup to arrive
// on an empty CU gets slot 1) -- a performance hint only, nothing depends on it for correctness.
#include "../vit_pytorch_amd/csrc/common.h"
#include "../vit_pytorch_amd/csrc/gemm_nt_plan.h"
#include "../vit_pytorch_amd/csrc/gemm_nt_epi.h"
#include <stdlib.h>
#include <type_traits>

#ifdef VITK_HALF_IS_F16
#define NTD_MFMA_ASM "v_mfma_f32_16x16x32_f16"
#else
#define NTD_MFMA_ASM "v_mfma_f32_16x16x32_bf16"
#endif

namespace {

constexpr int D_ATILE = 256 * 64;                   // activation rows of a K-step: 256 rows of 64 bytes
constexpr int D_WTILE = 128 * 64;                   // weight rows of a K-step: 128 rows of 64 bytes (half of a K-blocked block)
constexpr int D_WBLOCK = 256 * 64;                  // a K-blocked W block (vitk_pack_w_nt): 256 rows
constexpr int D_STAGE = D_ATILE + D_WTILE;          // 24 KiB
constexpr int D_RING = 3 * D_STAGE;                 // 72 KiB
constexpr int D_LDS = D_RING + 64;                  // + the phase word

#define D_PIN() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ int d_swz(int x) { return (0x1320 >> (4 * (x & 3))) & 3; }   // permutation [0,2,3,1] (the 8-wave kernel's)

// tile t of the grouped order: groups of gn 128-column tiles, n fastest inside a group
__device__ __forceinline__ void d_grouped_tile(int t, int tiles_m, int tiles_n, int gn, int& tm, int& tn) {
    const int per_group = gn * tiles_m;
    const int g = t / per_group;
    const int r = t - g * per_group;
    const int rem = tiles_n - g * gn;
    const int w = rem < gn ? rem : gn;
    tm = r / w;
    tn = g * gn + (r - tm * w);
}

struct NtdArgs {
    const char* A; long long lda;      // element strides; operands are 2-byte elements
    const char* W; long long ldw;      // ldw == 0: K-blocked (vitk_pack_w_nt)
    void* C; long long ldc;
    int M, N, K;
    const __bf16* bias; const void* resid; __bf16* aux; float* csum;
    int tiles_n, group_n, tiles_m, n_tiles, nt;      // 128-column tiles; FULL interior tiles only: rows [0, 256 tiles_m), N % 128 == 0
    int delay;          // late start of a CU's second workgroup, in s_sleep 16 (~1024 cycles)
    int dbg;            // experiments: bit 0 = skip the epilogue (main loop alone), bit 1 = phase by block index (second half of the grid) instead of
                        // the wave slot, bit 2 = every workgroup in phase (no late start)
    unsigned long long* stamps;        // experiments: per workgroup {hw id | xcc << 32, phase, start, end} (s_memrealtime), or null
};

template <int OFF> __device__ __forceinline__ bf16x8 d_rd(unsigned lds_addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(OFF) : "memory");
    return v;
}
// MFMA as asm with the accumulator pinned to AGPRs; the Z form writes A.B (C = 0): the first K-step of a tile
template <bool Z> __device__ __forceinline__ void d_mfma(f32x4& c, const bf16x8& a, const bf16x8& b) {
    if constexpr (Z) asm volatile("" NTD_MFMA_ASM " %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
    else asm volatile("" NTD_MFMA_ASM " %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void d_gload_bf16x4(bf16x4& d, const __bf16* p) { asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }

// ABL (experiments, tools/nt_probe.hip): bit 0 no LDS-DMA in the loop, bit 1 no fragment reads, bit 2 no MFMA
template <int EPI, int ABL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_ntd_kernel(const NtdArgs p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr bool F32OUT = (EPI == VITK_EPI_RESID);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- this workgroup's tiles: XCD x owns a contiguous run of the list; its workgroups take every L-th tile of it ----
    const int xcd = blockIdx.x & 7, l0 = blockIdx.x >> 3, L = gridDim.x >> 3;
    const int ms = (int)(((long long)xcd * p.n_tiles) >> 3), count = (int)(((long long)(xcd + 1) * p.n_tiles) >> 3) - ms;
    if (l0 >= count) return;
    auto decode = [&](int idx, int& m0, int& n0, int& mt) {
        int tn;
        d_grouped_tile(ms + idx, p.tiles_m, p.tiles_n, p.group_n, mt, tn);
        m0 = mt * 256; n0 = tn * 128;
    };

    // ---- which of the CU's two workgroups is this?  The late one sleeps before its first DMA piece. ----
    unsigned long long t_start = 0;
    unsigned hwid = 0, xcc = 0;
    {
        unsigned* ph = reinterpret_cast<unsigned*>(lds + D_RING);
        if (tid == 0) {
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned late = (p.dbg & 2) ? (blockIdx.x >= (gridDim.x >> 1) ? 1u : 0u) : (hwid & 1u);
            if (p.dbg & 4) late = 0;
            *ph = late;
        }
        __syncthreads();
        const unsigned late = *ph;
        if (p.stamps && tid == 0) t_start = __builtin_amdgcn_s_memrealtime();
        if (late) { for (int i = p.delay; i > 0; --i) __builtin_amdgcn_s_sleep(16); }
        if (p.stamps && tid == 0) {
            p.stamps[4 * blockIdx.x + 0] = (unsigned long long)hwid | ((unsigned long long)xcc << 32);
            p.stamps[4 * blockIdx.x + 1] = late;
            p.stamps[4 * blockIdx.x + 2] = t_start;
        }
    }

    // ---- producer: the LDS-DMA stream runs three K-steps ahead of the MFMAs, across tile boundaries ----
    // 64-byte rows, a wave instruction fills 16 rows.  Activation tile: wave w owns pieces 4w .. 4w + 3 (LDS row R = tile row R); the LDS image
    // is lane-linear, so the bank swizzle sits in the SOURCE offset: position s of LDS row R holds logical 16-byte chunk s ^ d_swz(R >> 2).
    // W tile: wave w owns pieces 2w, 2w + 1; LDS row R = 64 q + 16 fn + c holds W row 64 q + 4 c + fn of the 128-column tile.  K-blocked W
    // (ldw == 0) IS that image: block (256-column tile, K-step) after block, this tile's half (n0 / 128 & 1) of it.
    const int srow = lane >> 2, spos = lane & 3;
    const int schunk = spos ^ d_swz(lane >> 4);
    int avo[4], wvo[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) avo[j] = (int)(((long long)(64 * wave + 16 * j + srow) * p.lda + schunk * 8) * 2);
#pragma unroll
    for (int j = 0; j < 2; ++j)
        wvo[j] = p.ldw == 0 ? (2 * wave + j) * 1024 + lane * 16
                            : (int)(((long long)(64 * (wave >> 1) + 4 * srow + 2 * (wave & 1) + j) * p.ldw + schunk * 8) * 2);
    const int w_kstride = p.ldw == 0 ? D_WBLOCK : 64;     // bytes between consecutive K-steps
    __amdgpu_buffer_rsrc_t a_rs, w_rs;
    int a_so = 0, w_so = 0;                             // scalar byte offsets of the producer's K-step
    auto setup_src = [&](int idx) {
        int m0, n0, mt;
        decode(idx, m0, n0, mt);
        const long long abytes = (255LL * p.lda + p.K) * 2;       // up to the last element of the tile's last row
        a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (long long)m0 * p.lda * 2), 0, (int)abytes, 0x00020000);
        a_so = 0;
        if (p.ldw == 0) {
            w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (long long)(n0 >> 8) * p.nt * D_WBLOCK), 0, p.nt * D_WBLOCK, 0x00020000);
            w_so = ((n0 >> 7) & 1) * D_WTILE;
        } else {
            const long long wbytes = (127LL * p.ldw + p.K) * 2;
            w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (long long)n0 * p.ldw * 2), 0, (int)wbytes, 0x00020000);
            w_so = 0;
        }
    };
    // piece q of the producer's K-step into stage `stg`: q < 4 activation piece 4w + q, else W piece 2w + q - 4
    auto dma = [&](int stg, int q) __attribute__((always_inline)) {
        if (q < 4) {
            char* dst = lds + stg * D_STAGE + (wave * 4 + q) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (void __attribute__((address_space(3)))*)dst, 16, avo[q], a_so, 0, 0);
        } else {
            char* dst = lds + stg * D_STAGE + D_ATILE + (wave * 2 + (q - 4)) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (void __attribute__((address_space(3)))*)dst, 16, wvo[q - 4], w_so, 0, 0);
        }
    };
    // after the 6 pieces of a K-step.  The producer changes tile at a FIXED point of the consumer's tile (before its K-step nt - 3: the stream
    // runs three K-steps ahead).  Past the end of the tile list the descriptors have range 0: the pieces still count (the waits stay uniform) but
    // fetch nothing and write zeros into stages nobody reads.
    auto advance = [&]() __attribute__((always_inline)) { a_so += 64; w_so += w_kstride; };
    auto next_src = [&](int idx) __attribute__((always_inline)) {
        if (idx < count) setup_src(idx);
        else {
            a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0, 0x00020000);
            w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0, 0x00020000);
            a_so = 0; w_so = 0;
        }
    };

    // ---- consumer ----
    const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)lds);
    unsigned a_rd, w_rd;                                 // + f * 1024 / + fn * 1024, + stage
    {
        const int fi = lane & 15, fg = lane >> 4;
        const int fpos = fg ^ d_swz(fi >> 2);
        a_rd = lds_base + (wm * 128 + fi) * 64 + fpos * 16;
        w_rd = lds_base + D_ATILE + (wn * 64 + fi) * 64 + fpos * 16;
    }

    int stg = 0;                // K-step counter mod 3: the stage whose fragments are in registers (the DMA of this K-step refills it)
    bf16x8 xa[8], wa[4], xb[8], wb[4];
    f32x4 acc[4][8];            // acc[fn][f][j]: row 16 f + 4 fg + j, column 4 fi + fn of the wave tile

    // one group of a K-step: 4 MFMAs on the current fragments (activation fragment G x the four W fragments), the fragments of the next K-step
    // (groups 0..3: two activation fragments each, groups 4..7: one W fragment each), one DMA piece in six of the eight groups
#define D_GROUP(G, Z, XC, WC, XN, WN) do { \
        if constexpr (!(ABL & 4)) d_mfma<Z>(acc[0][(G)], XC[(G)], WC[0]); \
        if constexpr (!(ABL & 2)) { if constexpr ((G) < 4) XN[2 * (G)] = d_rd<(2 * (G)) * 1024>(rdA); else WN[(G) - 4] = d_rd<((G) - 4) * 1024>(rdW); } \
        if constexpr (!(ABL & 4)) d_mfma<Z>(acc[1][(G)], XC[(G)], WC[1]); \
        if constexpr (!(ABL & 2)) { if constexpr ((G) < 4) XN[2 * (G) + 1] = d_rd<(2 * (G) + 1) * 1024>(rdA); } \
        if constexpr (!(ABL & 4)) d_mfma<Z>(acc[2][(G)], XC[(G)], WC[2]); \
        if constexpr (!(ABL & 1) && ((G) & 3) != 0) { D_PIN(); dma(stg, (G) - 1 - ((G) >> 2)); D_PIN(); } \
        if constexpr (!(ABL & 4)) d_mfma<Z>(acc[3][(G)], XC[(G)], WC[3]); \
    } while (0)
    // one K-step t.  The counted wait: own pieces of K-step t + 2 landed (those of t + 3, issued in this K-step, fly)
#define D_STEP(Z, XC, WC, XN, WN) do { \
        const int nstg_ = stg == 2 ? 0 : stg + 1; \
        const unsigned soff_ = nstg_ * D_STAGE; \
        const unsigned rdA = a_rd + soff_, rdW = w_rd + soff_; \
        __builtin_amdgcn_s_setprio(1); \
        D_GROUP(0, Z, XC, WC, XN, WN); D_GROUP(1, Z, XC, WC, XN, WN); D_GROUP(2, Z, XC, WC, XN, WN); D_GROUP(3, Z, XC, WC, XN, WN); \
        D_GROUP(4, Z, XC, WC, XN, WN); D_GROUP(5, Z, XC, WC, XN, WN); D_GROUP(6, Z, XC, WC, XN, WN); D_GROUP(7, Z, XC, WC, XN, WN); \
        __builtin_amdgcn_s_setprio(0); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      /* the next K-step's fragments are in registers */ \
        if constexpr (!(ABL & 1)) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); \
        D_PIN(); \
        __builtin_amdgcn_s_barrier();           /* stage t + 2 visible to all, stage t + 1 read by all */ \
        D_PIN(); \
        if constexpr (!(ABL & 1)) advance(); \
        stg = nstg_; \
    } while (0)

    // ---- epilogue of one tile: registers -> global, full lines, stores not waited for ----
    // the other NT kernels', over the wave's ONE 64-column block: fragment row f = 4 output rows per lane.  Every tile is interior (the launch
    // takes full tiles only), so operand rows come by uncounted asm loads D rows ahead and are waited for by exact counts (gemm_nt_epi.h).
    auto epilogue = [&](int m0, int n0, int mt) __attribute__((always_inline)) {
        constexpr int NR = 8;
        if (p.dbg & 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("" :: "a"(acc[i][j]));
            return;
        }
        // lane / wave coordinates re-derived behind an opaque statement: hipcc otherwise hoists the epilogue's per-row index arithmetic
        // above the tile loop and carries it through the main loop
        int elane;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(elane));
        int fi = elane & 15, fg = elane >> 4, ewm = wm, ewn = wn;
        asm volatile("" : "+v"(fi), "+v"(fg), "+s"(ewm), "+s"(ewn));
        const int mrow0 = m0 + ewm * 128 + 4 * fg;              // + 16 f + j
        const int ncolw = n0 + ewn * 64;                        // first column of the wave tile
        const long long obase4 = (long long)mrow0 * p.ldc + ncolw + 4 * fi;      // element (row mrow0, the lane's 4 columns)
        f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (q_has_bias<EPI>()) {
            if (p.bias) {
                // (an ordinary load: the compiler's own wait for it also drains the next tile's three K-steps of DMA, issued 1-3 K-steps ago)
                const bf16x4 bb = *reinterpret_cast<const bf16x4*>(p.bias + ncolw + 4 * fi);
                b4 = f32x4{(float)bb[0], (float)bb[1], (float)bb[2], (float)bb[3]};
            }
        }
        if constexpr (F32OUT) {
            // lane: rows mrow0 + 16 f + j, 4 consecutive f32 columns: 16 lanes = 256 contiguous bytes of a row
            float* Cf = reinterpret_cast<float*>(p.C);
            const float* Rf = reinterpret_cast<const float*>(p.resid);
            constexpr int D = 2;
            f32x4 r[D][4];
            auto fetch = [&](int f, f32x4 (&dst)[4]) __attribute__((always_inline)) {
                long long orow = obase4 + (long long)(f * 16) * p.ldc;
                asm volatile("" : "+v"(orow));
                const float* rp = Rf + orow;
#pragma unroll
                for (int j = 0; j < 4; ++j) q_gload_f32x4(dst[j], rp + (long long)j * p.ldc);
            };
#pragma unroll
            for (int i = 0; i < D; ++i) fetch(i, r[i]);
            auto row = [&](auto rc) __attribute__((always_inline)) {
                constexpr int f = decltype(rc)::value;
                // (the AGPR -> VGPR copies of the row's accumulators are formed here, not behind the last MFMAs)
                D_PIN();
                asm volatile("" : "+a"(acc[0][f]), "+a"(acc[1][f]), "+a"(acc[2][f]), "+a"(acc[3][f]));
                f32x4 (&rr)[4] = r[f % D];
                q_wait_regs4<q_epi_younger(f, NR, D, 4)>(rr[0], rr[1], rr[2], rr[3]);
                long long ocp = obase4 + (long long)(f * 16) * p.ldc;
                asm volatile("" : "+v"(ocp));
                float* cp = Cf + ocp;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 v = f32x4{acc[0][f][j], acc[1][f][j], acc[2][f][j], acc[3][f][j]} + b4;
                    v += rr[j];
                    *reinterpret_cast<f32x4*>(cp + (long long)j * p.ldc) = v;
                }
                if constexpr (f + D < NR) fetch(f + D, rr);
            };
            row(std::integral_constant<int, 0>{}); row(std::integral_constant<int, 1>{}); row(std::integral_constant<int, 2>{}); row(std::integral_constant<int, 3>{});
            row(std::integral_constant<int, 4>{}); row(std::integral_constant<int, 5>{}); row(std::integral_constant<int, 6>{}); row(std::integral_constant<int, 7>{});
        } else if constexpr (EPI == VITK_EPI_RESID16) {
            // the residual epilogue with the stream in the 16-bit type: lane -> (row, 4 columns), 8-byte loads and stores (16 lanes = 128
            // contiguous bytes of a row); the sum is formed in f32 and rounded once
            __bf16* Cb = reinterpret_cast<__bf16*>(p.C);
            const __bf16* Rb = reinterpret_cast<const __bf16*>(p.resid);
            constexpr int D = 4;
            bf16x4 r[D][4];
            auto fetch = [&](int f, bf16x4 (&dst)[4]) __attribute__((always_inline)) {
                long long orow = obase4 + (long long)(f * 16) * p.ldc;
                asm volatile("" : "+v"(orow));
                const __bf16* rp = Rb + orow;
#pragma unroll
                for (int j = 0; j < 4; ++j) d_gload_bf16x4(dst[j], rp + (long long)j * p.ldc);
            };
#pragma unroll
            for (int i = 0; i < D; ++i) fetch(i, r[i]);
            auto row = [&](auto rc) __attribute__((always_inline)) {
                constexpr int f = decltype(rc)::value;
                D_PIN();
                asm volatile("" : "+a"(acc[0][f]), "+a"(acc[1][f]), "+a"(acc[2][f]), "+a"(acc[3][f]));
                bf16x4 (&rr)[4] = r[f % D];
                q_wait_regs4<q_epi_younger(f, NR, D, 4)>(rr[0], rr[1], rr[2], rr[3]);
                long long ocp = obase4 + (long long)(f * 16) * p.ldc;
                asm volatile("" : "+v"(ocp));
                __bf16* cp = Cb + ocp;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 v = f32x4{acc[0][f][j], acc[1][f][j], acc[2][f][j], acc[3][f][j]} + b4;
                    const bf16x4 q4 = rr[j];
                    v += f32x4{(float)q4[0], (float)q4[1], (float)q4[2], (float)q4[3]};
                    store4<__bf16>(cp + (long long)j * p.ldc, v);
                }
                if constexpr (f + D < NR) fetch(f + D, rr);
            };
            row(std::integral_constant<int, 0>{}); row(std::integral_constant<int, 1>{}); row(std::integral_constant<int, 2>{}); row(std::integral_constant<int, 3>{});
            row(std::integral_constant<int, 4>{}); row(std::integral_constant<int, 5>{}); row(std::integral_constant<int, 6>{}); row(std::integral_constant<int, 7>{});
        } else {
            // after the pair exchange: even lanes own row r = mrow0 + 16 f + 2 pr, odd lanes row r + 1, columns ncol8 .. + 7 of the block
            const int odd = fi & 1;
            __bf16* Cb = reinterpret_cast<__bf16*>(p.C);
            const long long obase = (long long)(mrow0 + odd) * p.ldc + ncolw + 8 * (fi >> 1);      // + 16 f ldc + 2 pr ldc
            constexpr bool AUX_IN = q_aux_in<EPI>();      // an (M, N) operand read in the epilogue (16-bit; MUL_AUX8: 8-bit codes)
            constexpr bool AUX8 = (EPI == VITK_EPI_MUL_AUX8);
            constexpr int DP = AUX8 ? 4 : 3;
            using HPre = std::conditional_t<AUX8, q_u32x2, bf16x8>;
            HPre hpre[DP][2];
            auto fetch_pre = [&](int f, HPre (&dst)[2]) __attribute__((always_inline)) {
                long long oa = obase + (long long)(f * 16) * p.ldc;
                asm volatile("" : "+v"(oa));        // (opaque: hipcc otherwise forms the addresses of all rows at the top of the epilogue)
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
            float cs[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) cs[e] = 0.f;
            auto frow = [&](auto rc) __attribute__((always_inline)) {
                constexpr int f = decltype(rc)::value;
                D_PIN();
                asm volatile("" : "+a"(acc[0][f]), "+a"(acc[1][f]), "+a"(acc[2][f]), "+a"(acc[3][f]));
                if constexpr (AUX_IN) q_wait_regs2<q_epi_younger(f, NR, DP, 2)>(hpre[f % DP][0], hpre[f % DP][1]);
                long long o0 = obase + (long long)(f * 16) * p.ldc;
                asm volatile("" : "+v"(o0));
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    // rows j0 = 2 pr and j0 + 1 of this lane's 4 columns, rounded to the 16-bit type
                    const int j0 = 2 * pr;
                    const unsigned a0 = q_pack2(acc[0][f][j0] + b4[0], acc[1][f][j0] + b4[1]);
                    const unsigned a1 = q_pack2(acc[2][f][j0] + b4[2], acc[3][f][j0] + b4[3]);
                    const unsigned c0 = q_pack2(acc[0][f][j0 + 1] + b4[0], acc[1][f][j0 + 1] + b4[1]);
                    const unsigned c1 = q_pack2(acc[2][f][j0 + 1] + b4[2], acc[3][f][j0 + 1] + b4[3]);
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
                        q_gelu_both8(q_widen8(v), gl, dgl);              // of the ROUNDED pre-activation, like BIAS_GELU
                        if constexpr (EPI == VITK_EPI_BIAS_GELU_DG8) *reinterpret_cast<q_u32x2*>(reinterpret_cast<unsigned char*>(p.aux) + o) = q_dg_encode8(dgl);
                        else *reinterpret_cast<bf16x8*>(p.aux + o) = q_narrow8(dgl);
                        *reinterpret_cast<bf16x8*>(Cb + o) = q_narrow8(gl);
                    } else if constexpr (AUX_IN) {
                        q_f32x8 fac;
                        if constexpr (AUX8) fac = q_dg_decode8(hpre[f % DP][pr]);
                        else if constexpr (EPI == VITK_EPI_MUL_AUX) fac = q_widen8(hpre[f % DP][pr]);
                        else fac = q_gelu_grad8(q_widen8(hpre[f % DP][pr]));
                        const q_f32x8 g = q_widen8(v) * fac;
                        const bf16x8 g8 = q_narrow8(g);
                        *reinterpret_cast<bf16x8*>(Cb + o) = g8;
#pragma unroll
                        for (int e = 0; e < 8; ++e) cs[e] += (float)g8[e];     // of the ROUNDED values: what colsum(C) would read
                    }
                }
                if constexpr (AUX_IN) {
                    // (the column sums are accumulated HERE: hipcc otherwise sinks the adds below the last row and keeps the whole tile's values alive)
                    asm volatile("" : "+v"(cs[0]), "+v"(cs[1]), "+v"(cs[2]), "+v"(cs[3]), "+v"(cs[4]), "+v"(cs[5]), "+v"(cs[6]), "+v"(cs[7]));
                    if constexpr (f + DP < NR) fetch_pre(f + DP, hpre[f % DP]);
                }
            };
            frow(std::integral_constant<int, 0>{}); frow(std::integral_constant<int, 1>{}); frow(std::integral_constant<int, 2>{}); frow(std::integral_constant<int, 3>{});
            frow(std::integral_constant<int, 4>{}); frow(std::integral_constant<int, 5>{}); frow(std::integral_constant<int, 6>{}); frow(std::integral_constant<int, 7>{});
            if constexpr (AUX_IN) {
                if (p.csum) {
                    // bias gradient by-product: the 8 lanes (c ^ 1, 4 row groups) that own the same 8 columns are summed in
                    // registers; one partial row per (m-tile, wm)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float v = cs[e];
                        v += __builtin_bit_cast(float, q_dpp_xor1(__builtin_bit_cast(unsigned, v)));
                        unsigned u = __builtin_bit_cast(unsigned, v);
                        auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
                        v = __builtin_bit_cast(float, (unsigned)a[0]) + __builtin_bit_cast(float, (unsigned)a[1]);
                        u = __builtin_bit_cast(unsigned, v);
                        auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                        cs[e] = __builtin_bit_cast(float, (unsigned)b[0]) + __builtin_bit_cast(float, (unsigned)b[1]);
                    }
                    if (fg == 0 && !odd) {
                        float* cp = p.csum + (long long)(2 * mt + ewm) * p.N + ncolw + 8 * (fi >> 1);
                        *reinterpret_cast<f32x4*>(cp) = f32x4{cs[0], cs[1], cs[2], cs[3]};
                        *reinterpret_cast<f32x4*>(cp + 4) = f32x4{cs[4], cs[5], cs[6], cs[7]};
                    }
                }
            }
        }
    };

    // ---- prologue: K-steps 0..2 in flight, 0 and 1 landed, the fragments of K-step 0 in registers ----
    setup_src(l0);
    if constexpr (!(ABL & 1)) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
#pragma unroll
            for (int q = 0; q < 6; ++q) dma(s, q);
            advance();
        }
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
    D_PIN();
    __builtin_amdgcn_s_barrier();
    D_PIN();
    {
        const unsigned rdA = a_rd, rdW = w_rd;
        xa[0] = d_rd<0 * 1024>(rdA); xa[1] = d_rd<1 * 1024>(rdA); xa[2] = d_rd<2 * 1024>(rdA); xa[3] = d_rd<3 * 1024>(rdA);
        xa[4] = d_rd<4 * 1024>(rdA); xa[5] = d_rd<5 * 1024>(rdA); xa[6] = d_rd<6 * 1024>(rdA); xa[7] = d_rd<7 * 1024>(rdA);
        wa[0] = d_rd<0 * 1024>(rdW); wa[1] = d_rd<1 * 1024>(rdW); wa[2] = d_rd<2 * 1024>(rdW); wa[3] = d_rd<3 * 1024>(rdW);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    D_PIN();
    __builtin_amdgcn_s_barrier();           // stage 0 has been read by everyone: K-step 0 may refill it
    D_PIN();
    if constexpr (ABL & 2) {
#pragma unroll
        for (int f = 0; f < 8; ++f) xb[f] = xa[f];
#pragma unroll
        for (int f = 0; f < 4; ++f) wb[f] = wa[f];
    }

    for (int idx = l0; idx < count; idx += L) {
        int m0, n0, mt;
        decode(idx, m0, n0, mt);
        // K-steps in pairs (the two fragment sets swap roles); the first K-step writes the accumulators with C = 0
        D_STEP(true, xa, wa, xb, wb);
        D_STEP(false, xb, wb, xa, wa);
        for (int kt = 2; kt + 4 < p.nt; kt += 2) {
            D_STEP(false, xa, wa, xb, wb);
            D_STEP(false, xb, wb, xa, wa);
        }
        D_STEP(false, xa, wa, xb, wb);
        next_src(idx + L);       // the pieces of the last three K-steps are the next tile's first three
        D_STEP(false, xb, wb, xa, wa);
        D_STEP(false, xa, wa, xb, wb);
        D_STEP(false, xb, wb, xa, wa);       // reads the NEXT tile's first fragments
        // the asm MFMAs' results are complete before the compiler's reads of them (it does not see the MFMAs' latency); nothing is
        // scheduled across the pin
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        D_PIN();
        epilogue(m0, n0, mt);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the surplus DMA pieces must not outlive the workgroup's LDS allocation
    if (p.stamps && tid == 0) p.stamps[4 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
#undef D_STEP
#undef D_GROUP
}

template <typename Kern>
int d_set_max_lds(Kern kernel, int bytes) {
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

}  // namespace

// Shapes the co-resident kernel takes: K-steps in pairs with a peeled first pair and four peeled last steps (K % 64 == 0, K >= 256), whole
// 256-column blocks of W (N % 256 == 0), 32-bit descriptor offsets.  Like the four-wave kernel it computes FULL 256-row tiles only, rows
// [0, 256 tiles_m); the caller gives the remaining rows to the 8-wave kernel.  Every epilogue but GELU_BWD (the pre-round-4 backward form, which
// does not fit 128 VGPRs beside the next tile's fragments without scratch; the four-wave kernel keeps it).
bool gemm_ntd_serves(int64_t M, int64_t N, int64_t K) {
    return !((K % 64) || K < 256 || (N % 256) || M < 256 || N > 16128 || (K / 32) * (long long)D_WBLOCK >= (1LL << 31));
}

// grid = resident workgroups = 2 per CU (a multiple of 16); delay < 0: the default late start (half a tile)
int gemm_ntd_launch(int tiles_m, int grid, const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                    int64_t N, int64_t K, int epilogue, const void* bias, const void* resid, void* aux, float* csum, int abl, int dbg, int delay,
                    void* stamps, void* stream) {
    NtdArgs a;
    a.A = (const char*)A; a.lda = lda; a.W = (const char*)W; a.ldw = ldw; a.C = C; a.ldc = ldc;
    a.M = 256 * tiles_m; a.N = (int)N; a.K = (int)K;
    a.bias = (const __bf16*)bias; a.resid = resid; a.aux = (__bf16*)aux; a.csum = csum;
    a.tiles_n = (int)(N / 128); a.tiles_m = tiles_m; a.n_tiles = a.tiles_m * a.tiles_n; a.nt = (int)(K / 32);
    // grouped tile order: the four-wave kernel's groups (4 n-tiles of 256 columns for K <= 768, balanced groups of <= 6 beyond), in 128-column tiles
    const int tn256 = (int)(N / 256);
    int g256 = tn256;
    if (tn256 > 8) g256 = K <= 768 ? 4 : (tn256 + (tn256 + 5) / 6 - 1) / ((tn256 + 5) / 6);
    if (vitk_exp("VITK_GROUP_N")) { const int g = atoi(vitk_exp("VITK_GROUP_N")); g256 = g > 0 && g < tn256 ? g : tn256; }
    a.group_n = 2 * g256;
    // the late start: half of (main loop alone + epilogue) of a tile; ~0.22 us per K-step of a 256 x 128 tile alone + ~3 us of epilogue
    a.delay = delay >= 0 ? delay : (int)((0.22 * (double)a.nt + 3.0) * 0.5 * 2.4);      // us -> units of ~1024 cycles at 2.4 GHz
    if (a.delay > 255) a.delay = 255;
    a.dbg = dbg;
    a.stamps = (unsigned long long*)stamps;
    hipStream_t st = (hipStream_t)stream;
    if (tiles_m <= 0 || grid < 16) VITK_FAIL(VITK_E_ARG, "gemm_nt_bf16 (duo): nothing to do");
#define NTD_LAUNCH1(E, AB) do { \
        static const int rc__ = d_set_max_lds(gemm_ntd_kernel<E, AB>, D_LDS); \
        if (rc__ != 0) VITK_FAIL(rc__, "gemm_nt_bf16 (duo): cannot enable %d B of LDS", D_LDS); \
        hipLaunchKernelGGL((gemm_ntd_kernel<E, AB>), dim3((unsigned)grid), dim3(256), D_LDS, st, a); \
    } while (0)
#ifdef NTD_PROBE
#define NTD_LAUNCH_ALL(E) do { \
        switch (abl) { \
            case 0: NTD_LAUNCH1(E, 0); break; case 1: NTD_LAUNCH1(E, 1); break; case 2: NTD_LAUNCH1(E, 2); break; case 3: NTD_LAUNCH1(E, 3); break; \
            case 4: NTD_LAUNCH1(E, 4); break; case 5: NTD_LAUNCH1(E, 5); break; case 6: NTD_LAUNCH1(E, 6); break; case 7: NTD_LAUNCH1(E, 7); break; \
            default: VITK_FAIL(VITK_E_ARG, "gemm_nt_bf16 (duo): bad ablation %d", abl); \
        } } while (0)
#define NTD_LAUNCH(E) do { if (abl == 0) NTD_LAUNCH1(E, 0); else if (abl == 7) NTD_LAUNCH1(E, 7); else VITK_FAIL(VITK_E_ARG, "gemm_nt_bf16 (duo): ablation %d exists for EPI_NONE only", abl); } while (0)
#else
#define NTD_LAUNCH(E) do { if (abl) VITK_FAIL(VITK_E_ARG, "gemm_nt_bf16 (duo): ablations exist in tools/nt_probe.hip only"); NTD_LAUNCH1(E, 0); } while (0)
#define NTD_LAUNCH_ALL(E) NTD_LAUNCH(E)
#endif
    switch (epilogue) {
        case VITK_EPI_NONE: NTD_LAUNCH_ALL(VITK_EPI_NONE); break;
#ifndef NTD_PROBE_LEAN
        case VITK_EPI_BIAS: NTD_LAUNCH(VITK_EPI_BIAS); break;
        case VITK_EPI_BIAS_GELU: NTD_LAUNCH(VITK_EPI_BIAS_GELU); break;
        case VITK_EPI_RESID: NTD_LAUNCH(VITK_EPI_RESID); break;
        case VITK_EPI_BIAS_GELU_DG: NTD_LAUNCH(VITK_EPI_BIAS_GELU_DG); break;
        case VITK_EPI_MUL_AUX: NTD_LAUNCH(VITK_EPI_MUL_AUX); break;
        case VITK_EPI_BIAS_GELU_DG8: NTD_LAUNCH(VITK_EPI_BIAS_GELU_DG8); break;
        case VITK_EPI_MUL_AUX8: NTD_LAUNCH(VITK_EPI_MUL_AUX8); break;
        case VITK_EPI_RESID16: NTD_LAUNCH(VITK_EPI_RESID16); break;
#endif
        default: VITK_FAIL(VITK_E_ARG, "gemm_nt_bf16 (duo): bad epilogue %d", epilogue);
    }
#undef NTD_LAUNCH_ALL
#undef NTD_LAUNCH
#undef NTD_LAUNCH1
    VITK_CHECK_LAUNCH("gemm_nt_bf16 (duo)");
    return 0;
}
