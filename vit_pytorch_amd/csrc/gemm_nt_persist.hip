// gemm_nt_persist.hip -- persistent NT GEMM for gfx950:  C[M,N] = A[M,K] . W[N,K]^T (+ fused epilogue)
//
// Replaces nn.Linear forward (vit.py:20,23,44,47) and the dX GEMMs of its autograd at the sizes that dominate the step
// (M >= 1024 token rows).  Same arithmetic as gemm_nt256pp_kernel (gemm_bf16.hip): 256-column tiles, 8 waves (2 x 4), wave
// tile 128 x 64 on v_mfma_f32_16x16x32, K-step 32, operands HBM -> LDS by global_load_lds into four 32 KiB stages with
// counted vmcnt, the two waves of a SIMD running one barrier slot apart (ping-pong).  What is different:
//
//  * ONE RESIDENT WORKGROUP PER CU walks a static list of tiles.  The K-step stream (tile, k) is continuous: the LDS-DMA of the
//    next tile's first three K-steps is issued during the last three K-steps of the current tile and stays in flight across
//    the epilogue, so a tile has no pipeline fill.  At K = 768 (24 K-steps) fill + drain used to be ~20 % of a tile.
//  * THE EPILOGUE NEVER TOUCHES THE DMA RING (so the ring keeps running under it).  Every store instruction covers FULL
//    128-byte lines (a first version with 64-byte pieces of 16 rows
//    per instruction measured 0.82-0.95x the per-tile kernel: the epilogue is store-ISSUE bound, cost ~ lines touched):
//    the MFMA "A" operand is the activation fragment and "B" the W fragment, so a lane holds output rows 4g..4g+3 of ONE
//    column; the LDS-DMA places W row  4c + fn  of the wave's 64 columns at LDS row  16 fn + c  (a per-lane SOURCE address,
//    free), so lane c's four fragments are 4 CONSECUTIVE columns 4c..4c+3 of each of its rows.  f32 outputs (residual
//    epilogue): 16 bytes per lane, 16 lanes = 256 contiguous bytes of a row, 4 rows per instruction.  16-bit outputs: lane
//    pairs (c, c^1) trade the halves of two rows through a DPP quad permute, after which even lanes own 8 consecutive
//    columns of row r and odd lanes of row r + 1: 8 lanes = one 128-byte line, 8 lines per dwordx4 instruction.
//  * the bias vector is staged ONCE per workgroup in the 32 KiB of LDS above the ring: a vector load in the epilogue would
//    queue behind the LDS-DMA stream on vmcnt (hipcc waits vmcnt(0) for it -- an exposed memory round trip per tile).
//  * TILE HEIGHTS 256 AND 128 in one launch (gemm_nt_plan.h): the rows that would form a mostly idle last round of 256-row
//    tiles are cut into 128-row tiles (wave tile 64 x 64, two barrier slots per K-step instead of four).
//  * the bias-gradient column sums of the GELU' epilogue are reduced in registers (DPP + v_permlane swaps).
//  * W may arrive K-BLOCKED (ldw == 0; vitk_pack_w_nt below, done once per weight value): what bounds the main loop is the
//    LDS-DMA feed, and the feed is bound by REQUESTS -- [measured, tools/ldrow.hip] a CU streams a 32 KiB K-step in 0.59 us as
//    64-byte row pieces (16 half lines per instruction) and in 0.34-0.38 us as full 128-byte lines, against 0.49 us of MFMA work.
//    With the W half of every K-step contiguous the eight ViT-B/16 GEMM shapes run 1.696 -> 1.585 ms (K = 3072: -10 %).
//
// Scheduling of slots per K-step (group A = waves 0-3, group B = waves 4-7 one slot behind):
//   256-row tile:  R0 (read W + X[0..3] fragments, DMA A of K-step +3)  M0 (16 MFMA)  R1 (read X[4..7], DMA W of +3, counted
//                  wait for K-step +1)  M1 (16 MFMA)
//   128-row tile:  R0 (read W + X[0..3], DMA A and W of +3, counted wait)  M0 (16 MFMA)
// LDS hazards are those of gemm_nt256pp_kernel: a stage is re-filled only after the barrier that follows the last read of it
// by BOTH groups, and read only after every wave's counted vmcnt plus a barrier.
#include "common.h"
#include "gemm_nt_plan.h"
#include "gemm_nt_epi.h"
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include <unordered_map>
#include <type_traits>

namespace {

constexpr int Q_TILE_BYTES = 256 * 64;              // one operand, one K-step: 256 rows of 64 bytes
constexpr int Q_STAGE_BYTES = 2 * Q_TILE_BYTES;     // 32 KiB
constexpr int Q_LDS_BYTES = 4 * Q_STAGE_BYTES;      // 128 KiB ring
constexpr int Q_LDS_MAX = Q_LDS_BYTES + 32768;      // + bias image
constexpr int Q_TK_LINE = 32;                       // words per 128-byte line: one counter per line
constexpr int Q_TK_SLOT_WORDS = 32 * Q_TK_LINE;     // 17 lines used: tickets [0, 8), per-XCD exits [8, 16), chip exit 16

#define QQ_BARRIER() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

__device__ __forceinline__ int q_swz(int x) { return (0x1320 >> (4 * (x & 3))) & 3; }   // permutation [0,2,3,1]

__device__ __forceinline__ void q_grouped_tile(int t, int tiles_m, int tiles_n, int gn, int& tm, int& tn) {
    const int per_group = gn * tiles_m;
    const int g = t / per_group;
    const int r = t - g * per_group;
    const int rem = tiles_n - g * gn;
    const int w = rem < gn ? rem : gn;
    tm = r / w;
    tn = g * gn + (r - tm * w);
}

struct NtpArgs {
    const char* A; long long lda;      // element strides; operands are 2-byte elements
    const char* W; long long ldw;
    void* C; long long ldc;
    int M, N, K;
    const __bf16* bias; const float* resid; __bf16* aux; float* csum;
    int tiles_n, group_n, tm_main, tail_tm, n_main, n_tail, nt;
    unsigned drop_t, drop_seed; float inv_keep;     // fused nn.Dropout (vit.py:22,24,48): threshold 0 = off
    unsigned drop_m0;   // row of the dropout pattern that row 0 of this launch is (a launch on the last rows of a split call)
    int tail_first;
    unsigned* tickets;  // DYNAMIC tile tickets: 8 per-XCD counters zeroed before the launch (null: static tile lists) -- see the kernel
    int dbg;            // experiments: bit 0 = skip the epilogue (main loop alone)
    long long* stamps;  // experiments: s_memtime after the phases of K-steps 8..39 of one workgroup (VITK_NTP_STAMPS)
};

// returning add on a device word through the SCALAR memory path: the old value lands in an SGPR (wave-uniform by construction) and is
// tracked by lgkmcnt -- unlike a vector atomic it neither enters the in-order vmcnt queue of the LDS-DMA ring nor makes hipcc wait
// vmcnt(0).  [tools/probe_satomic.hip: gfx950 executes it; 1024 concurrent adds return 0..1023 exactly once.]
__device__ __forceinline__ unsigned q_satomic_add(unsigned* p, unsigned v) {
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(p) : "memory");
    return v;
}

template <int N_> __device__ __forceinline__ void q_wait_vm() {
    if constexpr (N_ == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N_ == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N_ == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N_ == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else static_assert(N_ < 0, "unsupported vmcnt");
}


template <int EPI, bool DYN>
__global__ __launch_bounds__(512) void gemm_ntp_kernel(const NtpArgs p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr bool F32OUT = (EPI == VITK_EPI_RESID);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    // ---- this workgroup's tiles: XCD x owns a contiguous run of the main list and of the tail list; its workgroups take
    //      every L-th tile of the concatenation, so the tiles an XCD works on at one time are neighbours in the grouped order
    //      (shared activation panel in ONE L2).  Placement b % 8 is the observed one; only speed depends on it.
    //      tail_first: the 128-row tiles lead, which leaves the workgroups that own one half a tile out of phase with the
    //      others for the rest of the launch (their store bursts then fall into the others' main loops).
    const int xcd = blockIdx.x & 7, l0 = blockIdx.x >> 3, L = gridDim.x >> 3;
    // queue q = XCD q's share of the main list and of the tail list
    auto q_geom = [&](int q, int& ms_, int& cm_, int& ts_, int& ct_) {
        ms_ = (int)(((long long)q * p.n_main) >> 3); cm_ = (int)(((long long)(q + 1) * p.n_main) >> 3) - ms_;
        ts_ = (int)(((long long)q * p.n_tail) >> 3); ct_ = (int)(((long long)(q + 1) * p.n_tail) >> 3) - ts_;
    };
    int ms, cm, ts, ct;
    q_geom(xcd, ms, cm, ts, ct);
    const int count = cm + ct;
    // DYNAMIC TICKETS (K >= 256): instead of the static list l0, l0 + L, ... every workgroup draws its next tile from
    // its XCD's counter (and, once that queue is dry, from the other XCDs').  Why: each workgroup needs a whole CU (128+ KiB of
    // LDS, 512 x 256 registers).  If another kernel -- an RCCL collective of the data-parallel step -- holds c CUs, c workgroups
    // of a static launch start only when others have FINISHED their whole lists: a second round, up to 2x the launch time.  With
    // tickets the resident workgroups simply draw more tiles and latecomers find the queues dry: ~256 / (256 - c).  The order
    // within a queue is the grouped order of the static lists, so the tiles an XCD works on at one time still share panels in
    // its L2.  A ticket is drawn TWO tiles ahead (the LDS-DMA stream crosses tile boundaries, so the producer needs tile n + 1
    // three K-steps before tile n ends), by wave 0 right after its epilogue with a SCALAR atomic (q_satomic_add: a vector atomic
    // would wait vmcnt(0), i.e. for the wave's own store drain, before the next tile's first K-step: measured +1 ms per step),
    // and handed to the other waves through an LDS word.
    constexpr bool dyn = DYN;
    if (!dyn && l0 >= count) return;
    const int ntiles = dyn ? 0 : (count - l0 + L - 1) / L;
    const int total_steps = ntiles * p.nt;
    const int c_first = p.tail_first ? ct : cm;          // length of the leading run (static lists)
    int* const tk_slots = reinterpret_cast<int*>(lds + Q_LDS_BYTES + p.tiles_n * 512);     // 4 words above the bias image
    int my_q = xcd;
    // ticket = (queue << 24) | index within the queue, or -1 when every queue is dry.  Wave-uniform (scalar atomics).  `dry` stops a
    // workgroup from hammering empty queues again; tk_done counts the workgroups that are through with the counters (see the end).
    bool dry = false;
    auto pop_ticket = [&]() -> int {
        if (dry) return -1;
        int q = my_q;
        for (int tries = 0; tries < 8; ++tries) {
            int a, b, c, d;
            q_geom(q, a, b, c, d);
            const unsigned v = q_satomic_add(p.tickets + q * Q_TK_LINE, 1u);
            if ((int)v < b + d) { my_q = q; return (q << 24) | (int)v; }
            q = (q + 1) & 7;
        }
        dry = true;
        return -1;
    };
    // the LAST workgroup to leave resets the slot for its next user (64 launches later): no memset node in front of every launch.
    // Two levels (32 workgroups per XCD counter, then 8): 256 exits on one word would serialise for ~7 us at the end of the launch.
    auto leave = [&]() {
        if (wave == 0) {
            const unsigned per_xcd = (gridDim.x >> 3) + (((int)gridDim.x & 7) > xcd ? 1u : 0u);
            if (q_satomic_add(p.tickets + (8 + xcd) * Q_TK_LINE, 1u) == per_xcd - 1) {
                if (q_satomic_add(p.tickets + 16 * Q_TK_LINE, 1u) == 7u && lane < 17) p.tickets[lane * Q_TK_LINE] = 0;
            }
        }
    };

    // tile of a ticket (dynamic) or of a list position of this XCD (static: queue = xcd)
    auto decode = [&](int tk, int& m0, int& half, int& n0, int& mt) {
        int qms = ms, qcm = cm, qts = ts, qct = ct, idx = tk;
        if (dyn) { q_geom(tk >> 24, qms, qcm, qts, qct); idx = tk & 0xffffff; }
        // order within a queue: tail_first = 1: the 128-row tiles, then the 256-row tiles (static lists); 0: the other way round;
        // 2 (experiments): HALF of the 128-row tiles, the 256-row tiles, the other half
        int tm, tn;
        bool is_tail; int k;
        if (p.tail_first == 2) {
            const int h1 = (qct + 1) >> 1;
            if (idx < h1) { is_tail = true; k = idx; }
            else if (idx < h1 + qcm) { is_tail = false; k = idx - h1; }
            else { is_tail = true; k = idx - qcm; }
        } else {
            const int first = p.tail_first ? qct : qcm;
            const bool lead = idx < first;
            k = lead ? idx : idx - first;
            is_tail = lead == (p.tail_first != 0);
        }
        if (!is_tail) {
            q_grouped_tile(qms + k, p.tm_main, p.tiles_n, p.group_n, tm, tn);
            m0 = tm * 256; half = 0; mt = tm;
        } else {
            q_grouped_tile(qts + k, p.tail_tm, p.tiles_n, p.group_n, tm, tn);
            m0 = p.tm_main * 256 + tm * 128; half = 1; mt = p.tm_main + tm;
        }
        n0 = tn * 256;
    };
    (void)c_first;

    // ---- producer: the LDS-DMA stream runs three K-steps ahead of the consumer, across tile boundaries ----
    // 64-byte rows, a wave instruction fills 16 rows; wave w owns LDS row groups 2w, 2w+1 of each operand tile.  The LDS image
    // is lane-linear, so the bank swizzle sits in the SOURCE address: position s of LDS row R holds logical 16-byte chunk
    // s ^ q_swz(R >> 2).  Activation tile: LDS row R = tile row R.  W tile: LDS row R = 64 q + 16 fn + c holds W row
    // 64 q + 4 c + fn (see the header: a lane's four B fragments are then 4 consecutive output columns).
    const int srow = lane >> 2, spos = lane & 3;
    const int schunk = spos ^ q_swz(lane >> 4);
    const char* a_src[2];
    const char* w_src[2];
    int p_idx = l0, p_kt = 0, p_g = 0;      // p_idx: list position (static) or ticket (dynamic)
    int p_n = 0;                            // tiles the producer has started (dynamic: slot index of the next ticket)
    bool p_more = true;
    auto next_ticket = [&](int n) -> int {  // written by wave 0 at least one barrier earlier
        return __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int*>(tk_slots + (n & 3)));
    };
    auto setup_src = [&](int idx) {
        int m0, half, n0, mt;
        decode(idx, m0, half, n0, mt);
        int mlast = m0 + (half ? 128 : 256);
        mlast = (mlast < p.M ? mlast : p.M) - 1;            // rows past the tile (half tiles) or past M re-read its last row
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rg = wave * 2 + j;
            int ar = m0 + rg * 16 + srow; ar = ar < mlast ? ar : mlast;
            int wr = n0 + (rg >> 2) * 64 + 4 * srow + (rg & 3); wr = wr < p.N ? wr : p.N - 1;
            a_src[j] = p.A + (long long)ar * p.lda * 2 + schunk * 16;
            w_src[j] = p.W + (long long)wr * p.ldw * 2 + schunk * 16;
            // K-blocked W (ldw == 0, vitk_pack_w_nt): block (n-tile, K-step) IS the 16 KiB LDS image, so a DMA instruction reads one
            // contiguous KiB (8 full lines) instead of 16 half lines
            if (p.ldw == 0) w_src[j] = p.W + (long long)(n0 >> 8) * p.nt * Q_TILE_BYTES + rg * 1024 + lane * 16;
        }
    };
    const int w_kstride = p.ldw == 0 ? Q_TILE_BYTES : 64;      // bytes between consecutive K-steps of a W row group
    // the four DMA instructions of a K-step as straight-line code (their block also holds MFMAs and must stay one basic
    // block).  Issued on EVERY K-step: when the tile list is exhausted the producer stays on its last K-step and re-loads it
    // into a stage nobody reads again, which keeps the counted waits uniform (always vmcnt(8)); the kernel drains before it ends.
    auto issue4 = [&]() __attribute__((always_inline)) {
        char* base = lds + (p_g & 3) * Q_STAGE_BYTES + wave * 2048;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(a_src[j] + p_kt * 64),
                                             (void __attribute__((address_space(3)))*)(base + j * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(w_src[j] + p_kt * w_kstride),
                                             (void __attribute__((address_space(3)))*)(base + Q_TILE_BYTES + j * 1024), 16, 0, 0);
        }
    };
    auto advance4 = [&]() __attribute__((always_inline)) {
        ++p_g;
        if (!p_more) return;
        if (++p_kt == p.nt) {
            bool more;
            if (dyn) { p_idx = next_ticket(++p_n); more = p_idx >= 0; }
            else { p_idx += L; more = p_idx < count; }
            if (more) { p_kt = 0; setup_src(p_idx); }
            else { p_more = false; p_kt = p.nt - 1; }
        }
    };

    // ---- bias -> LDS (above the ring), once ----
    const char* bias_lds = lds + Q_LDS_BYTES;
    if constexpr (q_has_bias<EPI>()) {
        const int ncols = p.tiles_n * 256;
        for (int i = tid * 8; i < ncols; i += 512 * 8) {
            bf16x8 v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (p.bias && i < p.N) v = *reinterpret_cast<const bf16x8*>(p.bias + i);
            *reinterpret_cast<bf16x8*>(lds + Q_LDS_BYTES + i * 2) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // written before the prologue's barrier publishes it
    }

    // ---- consumer ----
    const int fi = lane & 15, fg = lane >> 4;
    const int fpos = fg ^ q_swz(fi >> 2);
    const int a_off8 = (wm * 128 + fi) * 64 + fpos * 16;                 // + f * 1024   (256-row tile)
    const int a_off4 = (wm * 64 + fi) * 64 + fpos * 16;                  // + f * 1024   (128-row tile)
    const int w_off = Q_TILE_BYTES + (wn * 64 + fi) * 64 + fpos * 16;    // + fn * 1024

    int c_g = 0;            // global K-step counter of this workgroup
    // Q_TIMELINE builds (tools/nt_timeline.py: -DQ_TIMELINE=1, a separate libvitk_tl.so): s_memtime stamps of ONE wave of one workgroup along
    // the software-pipelined loop -- K-step starts, epilogue start, epilogue stores issued -- kept in LDS above the bias image (ds_write:
    // the lgkmcnt domain, so the exact vmcnt counts of the DMA ring stay exact) and copied to p.stamps when the workgroup ends.
#ifdef Q_TIMELINE
    const bool tl_on = p.stamps && blockIdx.x == 137 && lane == 0 && wave == (p.dbg >> 8);
    unsigned long long* const tl = reinterpret_cast<unsigned long long*>(lds + Q_LDS_BYTES + 8192);
    int tl_n = 0;
#define Q_TL(tag) do { if (tl_on && tl_n < 2040) { tl[tl_n++] = ((unsigned long long)__builtin_readcyclecounter() << 4) | (unsigned)(tag); } } while (0)
#else
#define Q_TL(tag) do { } while (0)
#endif
    auto run_tile = [&](auto fmw_c, int m0, int n0, int mt) {
        constexpr int FMW = decltype(fmw_c)::value;      // m-fragments per wave: 8 (256-row tile) or 4 (128-row tile)
        const int a_off = FMW == 8 ? a_off8 : a_off4;
        f32x4 acc[4][FMW];                               // acc[fn][f][j]: row 16 f + 4 fg + j, column 4 fi + fn (of the wave tile)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < FMW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        {
            // ---- software-pipelined main loop: no R slots.  The ds_reads of the NEXT fragments and the DMA issue sit between the
            //      MFMAs of the wave (its partner on the SIMD fills the bubbles), one barrier B(g) per K-step: "stage g + 1 is
            //      visible to everyone and stage g has been read by everyone" -- after it the DMA of K-step g + 4 may refill stage g.
            //      [measured, s_memtime stamps] in the slot version a K-step costs ~2 R0 + 2 M: issuing 8 ds_read_b128 takes a
            //      group ~230 cycles and issuing its 8 DMA pieces ~180 (the TA accepts one 1 KiB piece per ~21 cycles), all on the
            //      critical path, against 16 MFMAs = 272.  Every M block below is ONE basic block (unconditional reads, DMA issue
            //      without branches -- see issue4) so that the sched_group_barriers can interleave it.
            bf16x8 f0w[4], f0x[4], f1w[4], f1x[4], xg[4];
            // [measured, profiles/r04_nt_timeline.log: ~1,400 cycles per tile go by here with idle matrix cores and idle DMA issue.  Tried twice in
            //  round 4 and reverted: carrying these fragments from the previous tile's last K-step (which has read them) across the epilogue, and
            //  reading them half-way through the epilogue into registers of accumulator rows already stored -- either way the kernel, which
            //  sits at 256 registers, takes 70-300 bytes of scratch per lane, and the scratch build faulted on the device.]
            {   // fragments of this tile's first K-step (exposed once per tile; its stage was made visible by the previous B)
                const char* base = lds + (c_g & 3) * Q_STAGE_BYTES;
#pragma unroll
                for (int f = 0; f < 4; ++f) f0w[f] = *reinterpret_cast<const bf16x8*>(base + w_off + f * 1024);
#pragma unroll
                for (int f = 0; f < 4; ++f) f0x[f] = *reinterpret_cast<const bf16x8*>(base + a_off + f * 1024);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            // one K-step: MFMAs on (wf, xf), the next K-step's first fragments are read into (wn_, xn_)
            auto kstep = [&](bf16x8 (&wf)[4], bf16x8 (&xf)[4], bf16x8 (&wn_)[4], bf16x8 (&xn_)[4]) __attribute__((always_inline)) {
                Q_TL(1);
                const char* base = lds + (c_g & 3) * Q_STAGE_BYTES;
                const char* nbase = lds + ((c_g + 1) & 3) * Q_STAGE_BYTES;
                if constexpr (FMW == 8) {
                    // ---- M0 + reads of this step's second half of the activation fragments ----
#pragma unroll
                    for (int f = 0; f < 4; ++f) xg[f] = *reinterpret_cast<const bf16x8*>(base + a_off + (4 + f) * 1024);
#pragma unroll
                    for (int fn = 0; fn < 4; ++fn)
#pragma unroll
                        for (int f = 0; f < 4; ++f)
                            acc[fn][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[f], wf[fn], acc[fn][f], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                    __builtin_amdgcn_sched_barrier(0);             // (the asm wait below must not drift up between the MFMAs)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    q_wait_vm<8>();                                // own DMA of K-step c_g + 1 landed (c_g + 2, c_g + 3 may fly)
                    QQ_BARRIER();                                  // B(c_g)
                    // ---- M1 + reads of the next K-step's first fragments + DMA of K-step c_g + 4 into the stage just freed ----
#pragma unroll
                    for (int f = 0; f < 4; ++f) wn_[f] = *reinterpret_cast<const bf16x8*>(nbase + w_off + f * 1024);
#pragma unroll
                    for (int f = 0; f < 4; ++f) xn_[f] = *reinterpret_cast<const bf16x8*>(nbase + a_off + f * 1024);
                    issue4();
#pragma unroll
                    for (int fn = 0; fn < 4; ++fn)
#pragma unroll
                        for (int f = 0; f < 4; ++f)
                            acc[fn][4 + f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xg[f], wf[fn], acc[fn][4 + f], 0, 0, 0);
                } else {
                    q_wait_vm<8>();
                    QQ_BARRIER();                                  // B(c_g): this step's fragments were read during the previous step
#pragma unroll
                    for (int f = 0; f < 4; ++f) wn_[f] = *reinterpret_cast<const bf16x8*>(nbase + w_off + f * 1024);
#pragma unroll
                    for (int f = 0; f < 4; ++f) xn_[f] = *reinterpret_cast<const bf16x8*>(nbase + a_off + f * 1024);
                    issue4();
#pragma unroll
                    for (int fn = 0; fn < 4; ++fn)
#pragma unroll
                        for (int f = 0; f < 4; ++f)
                            acc[fn][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[f], wf[fn], acc[fn][f], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {          // MFMA, read, MFMA, (DMA), ...: 16 MFMAs carry 8 ds_read_b128 and 4 DMA pieces
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i & 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                advance4();
                ++c_g;
            };
            int kt = 0;
            for (; kt + 1 < p.nt; kt += 2) { kstep(f0w, f0x, f1w, f1x); kstep(f1w, f1x, f0w, f0x); }
            if (kt < p.nt) kstep(f0w, f0x, f1w, f1x);
        }

        // ---- epilogue: registers -> global, full lines, stores not waited for ----
        const int mrow0 = m0 + wm * (16 * FMW) + 4 * fg;          // + 16 f + j
        if (p.dbg & 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < FMW; ++j) asm volatile("" :: "v"(acc[i][j]));
            return;
        }
        Q_TL(2);
        const bool interior = (m0 + 32 * FMW <= p.M) && (n0 + 256 <= p.N);
        auto body = [&](auto int_c) {
            constexpr bool INT = decltype(int_c)::value;
            const int ncol4 = n0 + wn * 64 + 4 * fi;               // this lane's 4 columns before any exchange
            f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (q_has_bias<EPI>()) {
                const bf16x4 bb = *reinterpret_cast<const bf16x4*>(bias_lds + ncol4 * 2);
                b4 = f32x4{(float)bb[0], (float)bb[1], (float)bb[2], (float)bb[3]};
            }
            if constexpr (F32OUT) {
                // lane: rows mrow0 + 16 f + j, 4 consecutive f32 columns: 16 lanes = 256 contiguous bytes of a row
                const bool colok = INT || ncol4 < p.N;
                float* Cf = reinterpret_cast<float*>(p.C);
                if constexpr (INT && Q_EPI_ASM_RESID) {
                    // interior tile: residual rows by uncounted asm loads, D fragment rows ahead, exact-count waits (see q_gload_f32x4)
                    constexpr int D = Q_EPI_DEPTH_RESID;
                    f32x4 r[D][4];
                    const float* rp = p.resid + (long long)mrow0 * p.ldc + ncol4;
                    auto fetch = [&](int f, f32x4 (&dst)[4]) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) q_gload_f32x4(dst[j], rp + (long long)(f * 16 + j) * p.ldc);
                    };
#pragma unroll
                    for (int f = 0; f < D; ++f) fetch(f, r[f]);
                    auto row = [&](auto fc) {
                        constexpr int f = decltype(fc)::value;
                        f32x4 (&rr)[4] = r[f % D];
                        q_wait_regs4<q_epi_younger(f, FMW, D, 4)>(rr[0], rr[1], rr[2], rr[3]);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int m = mrow0 + f * 16 + j;
                            f32x4 v = f32x4{acc[0][f][j], acc[1][f][j], acc[2][f][j], acc[3][f][j]} + b4;
                            if (p.drop_t) {       // nn.Dropout on the Linear output, before the residual add (vit.py:24,48 + :80-81)
                                const unsigned hrow = drop_row((unsigned)m + p.drop_m0, p.drop_seed);
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = drop_keep(hrow, (unsigned)(ncol4 + e), p.drop_t) ? v[e] * p.inv_keep : 0.f;
                            }
                            v += rr[j];
                            *reinterpret_cast<f32x4*>(Cf + (long long)m * p.ldc + ncol4) = v;
                        }
                        if constexpr (f + D < FMW) fetch(f + D, rr);
                    };
                    row(std::integral_constant<int, 0>{}); row(std::integral_constant<int, 1>{});
                    row(std::integral_constant<int, 2>{}); row(std::integral_constant<int, 3>{});
                    if constexpr (FMW == 8) {
                        row(std::integral_constant<int, 4>{}); row(std::integral_constant<int, 5>{});
                        row(std::integral_constant<int, 6>{}); row(std::integral_constant<int, 7>{});
                    }
                } else {
                    f32x4 r[2][4];
                    auto fetch = [&](int f, f32x4 (&dst)[4]) {
    #pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            int m = mrow0 + f * 16 + j;
                            if (!INT) m = m < p.M ? m : p.M - 1;
                            dst[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                            if (colok) dst[j] = *reinterpret_cast<const f32x4*>(p.resid + (long long)m * p.ldc + ncol4);
                        }
                    };
                    fetch(0, r[0]);
                    fetch(1, r[1]);
    #pragma unroll
                    for (int f = 0; f < FMW; ++f) {
    #pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int m = mrow0 + f * 16 + j;
                            f32x4 v = f32x4{acc[0][f][j], acc[1][f][j], acc[2][f][j], acc[3][f][j]} + b4;
                            if (p.drop_t) {       // nn.Dropout on the Linear output, before the residual add (vit.py:24,48 + :80-81)
                                const unsigned hrow = drop_row((unsigned)m + p.drop_m0, p.drop_seed);
    #pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = drop_keep(hrow, (unsigned)(ncol4 + e), p.drop_t) ? v[e] * p.inv_keep : 0.f;
                            }
                            v += r[f & 1][j];
                            if (INT || (m < p.M && colok)) *reinterpret_cast<f32x4*>(Cf + (long long)m * p.ldc + ncol4) = v;
                        }
                        if (f + 2 < FMW) fetch(f + 2, r[f & 1]);
                    }
                }
            } else if constexpr (EPI == VITK_EPI_RESID16) {
                // the f32-residual epilogue with the stream in the 16-bit type: same lane -> (row, 4 columns) map, 8-byte loads and
                // stores (16 lanes = 128 contiguous bytes of a row); the sum is formed in f32 and rounded once
                const bool colok = INT || ncol4 < p.N;
                __bf16* Cb = reinterpret_cast<__bf16*>(p.C);
                const __bf16* Rb = reinterpret_cast<const __bf16*>(p.resid);
                bf16x4 r[2][4];
                auto fetch = [&](int f, bf16x4 (&dst)[4]) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        int m = mrow0 + f * 16 + j;
                        if (!INT) m = m < p.M ? m : p.M - 1;
                        dst[j] = bf16x4{0, 0, 0, 0};
                        if (colok) dst[j] = *reinterpret_cast<const bf16x4*>(Rb + (long long)m * p.ldc + ncol4);
                    }
                };
                fetch(0, r[0]);
                fetch(1, r[1]);
#pragma unroll
                for (int f = 0; f < FMW; ++f) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int m = mrow0 + f * 16 + j;
                        f32x4 v = f32x4{acc[0][f][j], acc[1][f][j], acc[2][f][j], acc[3][f][j]} + b4;
                        const bf16x4 rr = r[f & 1][j];
                        v += f32x4{(float)rr[0], (float)rr[1], (float)rr[2], (float)rr[3]};
                        if (INT || (m < p.M && colok)) store4<__bf16>(Cb + (long long)m * p.ldc + ncol4, v);
                    }
                    if (f + 2 < FMW) fetch(f + 2, r[f & 1]);
                }
            } else {
                // after the pair exchange: even lanes own row r = mrow0 + 16 f + 2 pr, odd lanes row r + 1, columns ncol8 .. + 7
                const int odd = fi & 1;
                const int ncol8 = n0 + wn * 64 + 8 * (fi >> 1);
                const bool colok = INT || ncol8 < p.N;
                __bf16* Cb = reinterpret_cast<__bf16*>(p.C);
                // GELU_BWD: saved pre-activations, fetched DP fragment rows ahead.  Interior tiles: uncounted asm loads + exact-count
                // waits (see q_gload_f32x4); per fragment row 2 loads and 2 stores
                constexpr bool AUX_IN = q_aux_in<EPI>();      // an (M, N) operand read in the epilogue (16-bit; MUL_AUX8: 8-bit codes)
                constexpr bool AUX8 = (EPI == VITK_EPI_MUL_AUX8);
                constexpr bool ASM_PRE = AUX_IN && INT && Q_EPI_ASM_PRE;
                constexpr int DP = ASM_PRE ? Q_EPI_DEPTH_PRE : 2;
                using HPre = std::conditional_t<AUX8, q_u32x2, bf16x8>;
                HPre hpre[DP][2];
                auto fetch_pre = [&](int f, HPre (&dst)[2]) {
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        int m = mrow0 + f * 16 + 2 * pr + odd;
                        if constexpr (AUX8) {
                            const unsigned char* a8 = reinterpret_cast<const unsigned char*>(p.aux);
                            if constexpr (ASM_PRE) q_gload_u32x2(dst[pr], a8 + (long long)m * p.ldc + ncol8);
                            else {
                                if (!INT) m = m < p.M ? m : p.M - 1;
                                dst[pr] = q_u32x2{0u, 0u};
                                if (colok) dst[pr] = *reinterpret_cast<const q_u32x2*>(a8 + (long long)m * p.ldc + ncol8);
                            }
                        } else if constexpr (ASM_PRE) q_gload_bf16x8(dst[pr], p.aux + (long long)m * p.ldc + ncol8);
                        else {
                            if (!INT) m = m < p.M ? m : p.M - 1;
                            dst[pr] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                            if (colok) dst[pr] = *reinterpret_cast<const bf16x8*>(p.aux + (long long)m * p.ldc + ncol8);
                        }
                    }
                };
                if constexpr (AUX_IN) {
#pragma unroll
                    for (int f = 0; f < DP; ++f) fetch_pre(f, hpre[f]);
                }
                float cs[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) cs[e] = 0.f;
                auto frow = [&](auto fc) {
                    constexpr int f = decltype(fc)::value;
                    if constexpr (ASM_PRE) q_wait_regs2<q_epi_younger(f, FMW, DP, 2)>(hpre[f % DP][0], hpre[f % DP][1]);
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        // rows j0 = 2 pr and j0 + 1 of this lane's 4 columns, rounded to the 16-bit type
                        const int j0 = 2 * pr;
                        const unsigned a0 = q_pack2b<q_has_bias<EPI>()>(acc[0][f][j0], acc[1][f][j0], b4[0], b4[1]);
                        const unsigned a1 = q_pack2b<q_has_bias<EPI>()>(acc[2][f][j0], acc[3][f][j0], b4[2], b4[3]);
                        const unsigned c0 = q_pack2b<q_has_bias<EPI>()>(acc[0][f][j0 + 1], acc[1][f][j0 + 1], b4[0], b4[1]);
                        const unsigned c1 = q_pack2b<q_has_bias<EPI>()>(acc[2][f][j0 + 1], acc[3][f][j0 + 1], b4[2], b4[3]);
                        // even lane keeps row j0 and receives the neighbour's 4 columns of it; odd lane likewise for row j0 + 1
                        const unsigned r0 = q_dpp_xor1(odd ? a0 : c0), r1 = q_dpp_xor1(odd ? a1 : c1);
                        const unsigned k0 = odd ? c0 : a0, k1 = odd ? c1 : a1;
                        const q_u32x4 w4 = odd ? q_u32x4{r0, r1, k0, k1} : q_u32x4{k0, k1, r0, r1};
                        const bf16x8 v = __builtin_bit_cast(bf16x8, w4);
                        const int m = mrow0 + f * 16 + j0 + odd;
                        const bool ok = INT || (m < p.M && colok);
                        const long long o = (long long)m * p.ldc + ncol8;
                        if constexpr (EPI == VITK_EPI_NONE || EPI == VITK_EPI_BIAS) {
                            if (ok) *reinterpret_cast<bf16x8*>(Cb + o) = v;
                        } else if constexpr (EPI == VITK_EPI_BIAS_GELU) {
                            if (ok) *reinterpret_cast<bf16x8*>(p.aux + o) = v;
                            bf16x8 g8 = q_narrow8(q_gelu8(q_widen8(v)));     // of the ROUNDED pre-activation
                            if (p.drop_t) {       // nn.Dropout after the GELU (vit.py:22): the saved pre-activation stays undropped
                                const unsigned hrow = drop_row((unsigned)m + p.drop_m0, p.drop_seed);
#pragma unroll
                                for (int e = 0; e < 8; ++e)
                                    g8[e] = drop_keep(hrow, (unsigned)(ncol8 + e), p.drop_t) ? (__bf16)((float)g8[e] * p.inv_keep) : (__bf16)0.f;
                            }
                            if (ok) *reinterpret_cast<bf16x8*>(Cb + o) = g8;
                        } else if constexpr (EPI == VITK_EPI_BIAS_GELU_DG || EPI == VITK_EPI_BIAS_GELU_DG8) {
                            q_f32x8 gl, dgl;
                            q_gelu_both8(q_widen8(v), gl, dgl);              // of the ROUNDED pre-activation, like BIAS_GELU
                            if constexpr (EPI == VITK_EPI_BIAS_GELU_DG8) {
                                if (ok) *reinterpret_cast<q_u32x2*>(reinterpret_cast<unsigned char*>(p.aux) + o) = q_dg_encode8(dgl);
                            } else {
                                if (ok) *reinterpret_cast<bf16x8*>(p.aux + o) = q_narrow8(dgl);
                            }
                            if (ok) *reinterpret_cast<bf16x8*>(Cb + o) = q_narrow8(gl);
                        } else if constexpr (AUX_IN) {
                            q_f32x8 fac;
                            if constexpr (AUX8) fac = q_dg_decode8(hpre[f % DP][pr]);
                            else if constexpr (EPI == VITK_EPI_MUL_AUX) fac = q_widen8(hpre[f % DP][pr]);
                            else fac = q_gelu_grad8(q_widen8(hpre[f % DP][pr]));
                            q_f32x8 g = q_widen8(v) * fac;
                            if (p.drop_t) {     // factor of the forward's dropout(gelu(pre)) at (m, n): same decision, same 1 / (1 - p)
                                const unsigned hrow = drop_row((unsigned)m + p.drop_m0, p.drop_seed);
#pragma unroll
                                for (int e = 0; e < 8; ++e) g[e] *= drop_keep(hrow, (unsigned)(ncol8 + e), p.drop_t) ? p.inv_keep : 0.f;
                            }
                            const bf16x8 g8 = q_narrow8(g);
                            if (ok) {
                                *reinterpret_cast<bf16x8*>(Cb + o) = g8;
                                q_cs_add8(cs, g8);                      // of the ROUNDED values: what colsum(C) would read
                            }
                        }
                    }
                    if constexpr (AUX_IN) {
                        if constexpr (f + DP < FMW) fetch_pre(f + DP, hpre[f % DP]);
                    }
                };
                frow(std::integral_constant<int, 0>{}); frow(std::integral_constant<int, 1>{});
                frow(std::integral_constant<int, 2>{}); frow(std::integral_constant<int, 3>{});
                if constexpr (FMW == 8) {
                    frow(std::integral_constant<int, 4>{}); frow(std::integral_constant<int, 5>{});
                    frow(std::integral_constant<int, 6>{}); frow(std::integral_constant<int, 7>{});
                }
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
                        if (fg == 0 && !odd && colok) {
                            float* cp = p.csum + (long long)(2 * mt + wm) * p.N + ncol8;
                            *reinterpret_cast<f32x4*>(cp) = f32x4{cs[0], cs[1], cs[2], cs[3]};
                            *reinterpret_cast<f32x4*>(cp + 4) = f32x4{cs[4], cs[5], cs[6], cs[7]};
                        }
                    }
                }
            }
        };
        if (interior) body(std::integral_constant<bool, true>{});
        else body(std::integral_constant<bool, false>{});
        Q_TL(3);
    };

    // ---- dynamic tickets: the first two, drawn by wave 0 ----
    if (dyn) {
        if (wave == 0) {
            // the first two tickets with ONE add on the own queue where both exist (the common case)
            int t0, t1;
            const unsigned v = q_satomic_add(p.tickets + xcd * Q_TK_LINE, 2u);
            if ((int)v + 1 < count) { t0 = (xcd << 24) | (int)v; t1 = t0 + 1; }
            else if ((int)v < count) { t0 = (xcd << 24) | (int)v; t1 = pop_ticket(); }
            else { t0 = pop_ticket(); t1 = t0 >= 0 ? pop_ticket() : -1; }
            if (lane == 0) { tk_slots[0] = t0; tk_slots[1] = t1; }
        }
        __syncthreads();           // nothing is in flight yet
        p_idx = next_ticket(0);
        if (p_idx < 0) { leave(); return; }     // every queue was dry before this workgroup started (a latecomer beside another kernel)
    }
    // ---- prologue: K-steps 0..2 in flight, K-step 0 landed ----
    setup_src(p_idx);
    // four stages in flight: B(g) frees stage g half a step before its next use
    issue4(); advance4(); issue4(); advance4(); issue4(); advance4(); issue4(); advance4();
    q_wait_vm<12>();
    QQ_BARRIER();              // also publishes the bias image

    if (dyn) {
        int n = 0;
        for (int cur = next_ticket(0); cur >= 0; cur = next_ticket(++n)) {
            int m0, half, n0, mt;
            decode(cur, m0, half, n0, mt);
            if (half) run_tile(std::integral_constant<int, 4>{}, m0, n0, mt);
            else run_tile(std::integral_constant<int, 8>{}, m0, n0, mt);
            if (wave == 0) {       // the ticket of tile n + 2: ~1 us of scalar atomic that falls into the store drain every wave is
                const int t = pop_ticket();      // about to sit out at its first counted wait of the next tile
                if (lane == 0) tk_slots[(n + 2) & 3] = t;
            }
        }
        leave();
    } else {
        for (int idx = l0; idx < count; idx += L) {
            int m0, half, n0, mt;
            decode(idx, m0, half, n0, mt);
            if (half) run_tile(std::integral_constant<int, 4>{}, m0, n0, mt);
            else run_tile(std::integral_constant<int, 8>{}, m0, n0, mt);
        }
    }
    q_wait_vm<0>();  // the surplus DMAs of the last K-steps must not outlive the workgroup's LDS allocation
#ifdef Q_TIMELINE
    if (tl_on) {
        p.stamps[0] = tl_n;
        for (int i = 0; i < tl_n; ++i) p.stamps[1 + i] = (long long)tl[i];
    }
#endif
}

// ---- dynamic tile tickets: 64 launch slots; a slot = 8 per-XCD ticket counters + 8 per-XCD exit counters + 1 chip exit counter,
// EACH ON ITS OWN 128-byte line ([measured] with the nine words in one 64-byte line every launch cost a fixed +14 us: atomics on
// one line serialise at ~27 ns each and 256 workgroups draw two tickets at once in the prologue); a launch leaves its slot zeroed
__device__ unsigned q_ticket_pool[64 * Q_TK_SLOT_WORDS];
unsigned* q_ticket_slot(hipStream_t st) {
    // the symbol has one address PER DEVICE the module is loaded on: resolved for the current device (one process per GPU is the deployment;
    // a single process driving several GPUs gets each device's own pool)
    static std::mutex mu;
    static unsigned* bases[64] = {};
    static std::atomic<unsigned> seq{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    unsigned* base;
    {
        std::lock_guard<std::mutex> g(mu);
        if (!bases[dev]) {
            void* ptr = nullptr;
            if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(q_ticket_pool)) != hipSuccess) return nullptr;
            bases[dev] = (unsigned*)ptr;
        }
        base = bases[dev];
    }
    (void)st;
    return base + (seq.fetch_add(1) & 63) * Q_TK_SLOT_WORDS;      // zero-initialised; every launch's last workgroup leaves its slot zeroed again
}

template <typename Kern>
int q_set_max_lds(Kern kernel, int bytes) {
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

int q_num_cus() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 8) return 256;
        return v;
    }();
    return n;
}

// makespan of the static assignment (see the kernel): XCD x owns main tiles [x n_main / 8, (x+1) n_main / 8) and the same share
// of the tail tiles; its L workgroups take every L-th tile of (main ++ tail).
long long q_makespan(int n_main, int n_tail, int L, long long c_full, long long c_half) {
    long long worst = 0;
    for (int x = 0; x < 8; ++x) {
        const int cm = (int)(((long long)(x + 1) * n_main) >> 3) - (int)(((long long)x * n_main) >> 3);
        const int ct = (int)(((long long)(x + 1) * n_tail) >> 3) - (int)(((long long)x * n_tail) >> 3);
        const int cnt = cm + ct;
        for (int l = 0; l < L && l < cnt; ++l) {
            const int nall = (cnt - l + L - 1) / L;
            const int nm = l < cm ? (cm - l + L - 1) / L : 0;
            const long long c = nm * c_full + (nall - nm) * c_half;
            if (c > worst) worst = c;
        }
    }
    return worst;
}

}  // namespace

NtpPlan ntp_plan(int64_t M, int64_t N, int64_t K, int64_t ldc, const void* aux) {
    NtpPlan pl{};
    pl.ok = (K % 32 == 0) && M >= 1024 && N >= 256 && (N % 8 == 0) && (ldc % 8 == 0) && (!aux || aligned16(aux)) &&
            M < (1 << 30) && N <= 16128 /* bias image + ticket words: 32 KiB of LDS above the ring */ && !vitk_exp("VITK_NO_PERSIST");
    if (!pl.ok) return pl;
    struct Key { int64_t m, n, k; bool operator==(const Key& o) const { return m == o.m && n == o.n && k == o.k; } };
    struct KeyHash { size_t operator()(const Key& k) const { return (size_t)(k.m * 1000003 + k.n * 10007 + k.k); } };
    static std::mutex mu;
    static std::unordered_map<Key, NtpPlan, KeyHash> cache;
    const Key key{M, N, K};
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = cache.find(key);
        if (it != cache.end()) return it->second;
    }
    pl.tiles_n = (int)((N + 255) / 256);
    pl.group_n = pl.tiles_n;
    if (pl.tiles_n > 8) pl.group_n = (pl.tiles_n + (pl.tiles_n + 5) / 6 - 1) / ((pl.tiles_n + 5) / 6);
    if (vitk_exp("VITK_GROUP_N")) pl.group_n = atoi(vitk_exp("VITK_GROUP_N")) > 0 ? atoi(vitk_exp("VITK_GROUP_N")) : pl.tiles_n;
    if (pl.group_n > pl.tiles_n) pl.group_n = pl.tiles_n;
    pl.nt = (int)(K / 32);
    pl.grid = q_num_cus() / 8 * 8;
    const int L = pl.grid / 8;
    // cost of a tile in barrier slots: 4 (2) per K-step of a 256-row (128-row) tile plus the epilogue
    const long long e_full = vitk_exp("VITK_NTP_EFULL") ? atoll(vitk_exp("VITK_NTP_EFULL")) : 16;
    const long long e_half = vitk_exp("VITK_NTP_EHALF") ? atoll(vitk_exp("VITK_NTP_EHALF")) : 12;
    const long long c_full = 4LL * pl.nt + e_full, c_half = 2LL * pl.nt + e_half;
    const int full_tm = (int)(M / 256);
    // candidate 0: no tail (every tile 256 rows, the last m-tile partial)
    int best_tm = (int)((M + 255) / 256), best_tail = 0;
    long long best = q_makespan(best_tm * pl.tiles_n, 0, L, c_full, c_half);
    const int dmax = full_tm < 512 ? full_tm : 512;
    for (int d = 0; d <= dmax; ++d) {
        const int tm_main = full_tm - d;
        const int64_t rows_tail = M - 256LL * tm_main;
        if (rows_tail <= 0) continue;
        const int tail_tm = (int)((rows_tail + 127) / 128);
        const long long c = q_makespan(tm_main * pl.tiles_n, tail_tm * pl.tiles_n, L, c_full, c_half);
        if (c < best) { best = c; best_tm = tm_main; best_tail = tail_tm; }
    }
    if (vitk_exp("VITK_NTP_TAIL")) {      // experiments: number of 256-row m-tiles moved to the tail
        const int d = atoi(vitk_exp("VITK_NTP_TAIL"));
        if (d < 0) { best_tm = (int)((M + 255) / 256); best_tail = 0; }
        else {
            best_tm = full_tm - d < 0 ? 0 : full_tm - d;
            best_tail = (int)((M - 256LL * best_tm + 127) / 128);
        }
    }
    pl.tm_main = best_tm;
    pl.tail_tm = best_tail;
    pl.n_main = pl.tm_main * pl.tiles_n;
    pl.n_tail = pl.tail_tm * pl.tiles_n;
    {
        std::lock_guard<std::mutex> g(mu);
        cache.emplace(key, pl);
    }
    return pl;
}

int gemm_ntp_launch(const NtpPlan& pl, const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M,
                    int64_t N, int64_t K, int epilogue, const void* bias, const float* resid, void* aux, float* csum, unsigned drop_t,
                    unsigned drop_seed, float inv_keep, void* stream, unsigned drop_m0) {
    NtpArgs a;
    a.drop_t = drop_t; a.drop_seed = drop_seed; a.inv_keep = inv_keep; a.drop_m0 = drop_m0;
    a.A = (const char*)A; a.lda = lda; a.W = (const char*)W; a.ldw = ldw; a.C = C; a.ldc = ldc;
    a.M = (int)M; a.N = (int)N; a.K = (int)K;
    a.bias = (const __bf16*)bias; a.resid = resid; a.aux = (__bf16*)aux; a.csum = csum;
    a.tiles_n = pl.tiles_n; a.group_n = pl.group_n; a.tm_main = pl.tm_main; a.tail_tm = pl.tail_tm;
    a.n_main = pl.n_main; a.n_tail = pl.n_tail; a.nt = pl.nt;
    // (static lists; with tickets the 128-row tiles go LAST, see below).  Tail FIRST de-phases the workgroups that own a half tile for the
    // rest of the launch, tail LAST keeps the sharers of an activation panel in step: [measured, profiles/r04_nt_tile_order_knobs_*] the FF1
    // GEMM (N = 3072: twelve n-tiles per panel, two 16-bit outputs) fetches 373 instead of 530 MB with the tail last and runs 3.97 instead of
    // 4.12 ms per step; every other epilogue is level or slightly better tail-first.  VITK_NTP_TAIL_LAST = 1 / 0 forces one order for all.
    static const int tail_env = vitk_exp("VITK_NTP_TAIL_LAST") ? (atoi(vitk_exp("VITK_NTP_TAIL_LAST")) ? 0 : 1) : -1;
    a.tail_first = tail_env >= 0 ? tail_env : ((epilogue == VITK_EPI_BIAS_GELU || epilogue == VITK_EPI_BIAS_GELU_DG || epilogue == VITK_EPI_BIAS_GELU_DG8) ? 0 : 1);
    // dynamic tile tickets (see the kernel): K >= 256 so that a ticket drawn two tiles ahead is always there in time.  WHEN: while
    // other kernels are expected on the chip -- vitk_set_cu_reserve(c > 0), which parallel.FlatGradSink opens around every
    // in-backward all-reduce -- or VITK_NTP_DYNAMIC=1.  Not by default: [measured, ViT-B/16 batch 256, no other kernel on the
    // chip] tickets cost the eight NT GEMMs of a layer 1.632 -> 1.664 ms (the N = 768 shapes, 2.6 tiles per workgroup, lose 6-18 us
    // each to the static plan's makespan-optimal lists; FF1 gains 14 us) and the training step 0.8 ms; with 16-64 CUs held by
    // another kernel they save 19-25 % (tools/cu_contention.py).  VITK_NTP_STATIC=1 forces the static lists.
    const char* st_env = vitk_switch("VITK_NTP_STATIC");
    const char* dy_env = vitk_switch("VITK_NTP_DYNAMIC");
    const bool want_dyn = (dy_env && dy_env[0] == '1') || vitk_get_cu_reserve() > 0;
    a.tickets = (pl.nt >= 8 && want_dyn && !(st_env && st_env[0] == '1')) ? q_ticket_slot((hipStream_t)stream) : nullptr;
    // queue order under tickets: the 256-row tiles first, the 128-row tiles last (longest-processing-time rule; [measured] sums of the
    // eight shapes: 1.664 ms vs 1.73 ms with the static lists' tail-first order or a half-and-half split).  VITK_NTP_DYN_ORDER = 0 / 1 / 2
    if (a.tickets) a.tail_first = vitk_exp("VITK_NTP_DYN_ORDER") ? atoi(vitk_exp("VITK_NTP_DYN_ORDER")) : 0;
    a.dbg = vitk_exp("VITK_NTP_DBG") ? atoi(vitk_exp("VITK_NTP_DBG")) : 0;
    a.stamps = vitk_exp("VITK_NTP_STAMPS") ? (long long*)strtoull(vitk_exp("VITK_NTP_STAMPS"), nullptr, 0) : nullptr;
#ifdef Q_TIMELINE
    const int lds_bytes = Q_LDS_MAX;
#else
    const int lds_bytes = Q_LDS_BYTES + pl.tiles_n * 512 + 16;      // ring + bias image (tiles_n * 256 columns of 2 bytes) + 4 ticket words
#endif
    hipStream_t st = (hipStream_t)stream;
#define NTP_LAUNCH1(E, Dn) do { \
            static const int rc__ = q_set_max_lds(gemm_ntp_kernel<E, Dn>, Q_LDS_MAX); \
            if (rc__ != 0) VITK_FAIL(rc__, "gemm_nt_bf16: cannot enable %d B of LDS", Q_LDS_MAX); \
            hipLaunchKernelGGL((gemm_ntp_kernel<E, Dn>), dim3((unsigned)pl.grid), dim3(512), lds_bytes, st, a); \
        } while (0)
#define NTP_LAUNCH(E) do { \
        if (a.tickets) NTP_LAUNCH1(E, true); \
        else NTP_LAUNCH1(E, false); \
    } while (0)
    switch (epilogue) {
        case VITK_EPI_NONE: NTP_LAUNCH(VITK_EPI_NONE); break;
        case VITK_EPI_BIAS: NTP_LAUNCH(VITK_EPI_BIAS); break;
        case VITK_EPI_BIAS_GELU: NTP_LAUNCH(VITK_EPI_BIAS_GELU); break;
        case VITK_EPI_RESID: NTP_LAUNCH(VITK_EPI_RESID); break;
        case VITK_EPI_GELU_BWD: NTP_LAUNCH(VITK_EPI_GELU_BWD); break;
        case VITK_EPI_BIAS_GELU_DG: NTP_LAUNCH(VITK_EPI_BIAS_GELU_DG); break;
        case VITK_EPI_MUL_AUX: NTP_LAUNCH(VITK_EPI_MUL_AUX); break;
        case VITK_EPI_BIAS_GELU_DG8: NTP_LAUNCH(VITK_EPI_BIAS_GELU_DG8); break;
        case VITK_EPI_MUL_AUX8: NTP_LAUNCH(VITK_EPI_MUL_AUX8); break;
        case VITK_EPI_RESID16: NTP_LAUNCH(VITK_EPI_RESID16); break;
        default: VITK_FAIL(VITK_E_ARG, "gemm_nt_bf16: bad epilogue %d", epilogue);
    }
#undef NTP_LAUNCH
#undef NTP_LAUNCH1
    VITK_CHECK_LAUNCH("gemm_nt_bf16 (persistent)");
    return 0;
}

// ---- K-blocked weights ---------------------------------------------------------------------------------------------------
// vitk_pack_w_nt: W (N x K, row-major) -> for every (256-column tile tn, K-step kt) the 16 KiB image the persistent kernel's
// LDS stage holds for it, blocks ordered (tn, kt): LDS row R = 64 q + 16 fn + c (64 bytes) is W row 256 tn + 64 q + 4 c + fn
// (rows past N repeat row N - 1), and position s of the row holds its 16-byte chunk s ^ q_swz(R >> 2) of K-step kt.
// The transposed flavour packs W^T (K x N) -- the operand of dX = dY . W -- straight from W.
namespace {
// logical operand: rows = TR ? K : N ("n"), reduction = TR ? N : K ("k")
__device__ __forceinline__ void pack_w_chunk(const __bf16* __restrict__ W, long long ldw, int N, int K, char* __restrict__ out, bool TR, long long chunk) {
    const int rows = TR ? K : N, red = TR ? N : K;
    const int nt = red >> 5;
    const long long total = (long long)((rows + 255) >> 8) * nt * 1024;
    if (chunk >= total) return;
    const int spos = (int)(chunk & 3), R = (int)((chunk >> 2) & 255);
    const long long blk = chunk >> 10;
    const int kt = (int)(blk % nt), tn = (int)(blk / nt);
    int n = tn * 256 + (R >> 6) * 64 + 4 * (R & 15) + ((R >> 4) & 3);
    n = n < rows ? n : rows - 1;
    const int k0 = kt * 32 + 8 * (spos ^ q_swz(R >> 2));
    bf16x8 v;
    if (!TR) v = *reinterpret_cast<const bf16x8*>(W + (long long)n * ldw + k0);
    else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = W[(long long)(k0 + e) * ldw + n];
    }
    *reinterpret_cast<bf16x8*>(out + chunk * 16) = v;
}
template <bool TR>
__global__ __launch_bounds__(256) void pack_w_kernel(const __bf16* __restrict__ W, long long ldw, int N, int K, char* __restrict__ out) {
    pack_w_chunk(W, ldw, N, K, out, TR, (long long)blockIdx.x * 256 + threadIdx.x);       // one 16-byte chunk per thread
}
}  // namespace

namespace {
// the same for a TABLE of weights in one launch (a training step re-packs every weight of the stack: 96 launches of ~5 us at ViT-B/16)
constexpr int PM_MAX = 96;
struct PackMany { const __bf16* W[PM_MAX]; char* out[PM_MAX]; int ldw[PM_MAX], N[PM_MAX], K[PM_MAX]; int blk0[PM_MAX + 1]; int count; unsigned tr[PM_MAX / 32]; };
__global__ __launch_bounds__(256) void pack_w_many_kernel(PackMany a) {
    int lo = 0, hi = a.count;                  // uniform binary search: blk0[lo] <= blockIdx.x < blk0[lo + 1]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int)blockIdx.x >= a.blk0[mid]) lo = mid; else hi = mid; }
    const bool tr = (a.tr[lo >> 5] >> (lo & 31)) & 1u;
    pack_w_chunk(a.W[lo], a.ldw[lo], a.N[lo], a.K[lo], a.out[lo], tr, (long long)((int)blockIdx.x - a.blk0[lo]) * 256 + threadIdx.x);
}
}  // namespace

extern "C" int64_t vitk_pack_w_nt_bytes(int64_t rows, int64_t red) {
    if (rows <= 0 || red <= 0 || (red & 31)) return 0;
    return ((rows + 255) / 256) * 256 * red * 2;
}

extern "C" int vitk_pack_w_nt(const void* W, int64_t ldw, int64_t N, int64_t K, void* out, void* out_t, void* stream) {
    if (!W || (!out && !out_t)) VITK_FAIL(VITK_E_ARG, "pack_w_nt: null pointer");
    if (N <= 0 || K <= 0 || ldw < K || (ldw & 7) || N > (1 << 24) || K > (1 << 24)) VITK_FAIL(VITK_E_SHAPE, "pack_w_nt: bad shape N=%lld K=%lld ldw=%lld", (long long)N, (long long)K, (long long)ldw);
    if ((out && (K & 31)) || (out_t && (N & 31))) VITK_FAIL(VITK_E_SHAPE, "pack_w_nt: the reduction dimension must be a multiple of 32");
    if (!aligned16(W) || (out && !aligned16(out)) || (out_t && !aligned16(out_t))) VITK_FAIL(VITK_E_ALIGN, "pack_w_nt: 16-byte aligned pointers required");
    hipStream_t st = (hipStream_t)stream;
    if (out) {
        const long long chunks = vitk_pack_w_nt_bytes(N, K) / 16;
        hipLaunchKernelGGL((pack_w_kernel<false>), dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, st, (const __bf16*)W, (long long)ldw, (int)N, (int)K, (char*)out);
    }
    if (out_t) {
        const long long chunks = vitk_pack_w_nt_bytes(K, N) / 16;
        hipLaunchKernelGGL((pack_w_kernel<true>), dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, st, (const __bf16*)W, (long long)ldw, (int)N, (int)K, (char*)out_t);
    }
    VITK_CHECK_LAUNCH("pack_w_nt");
    return 0;
}

extern "C" int vitk_pack_w_nt_many(const void* const* W, const int64_t* ldw, const int64_t* N, const int64_t* K, void* const* out,
                                   void* const* out_t, int64_t count, void* stream) {
    if (count <= 0) return 0;
    if (!W || !ldw || !N || !K || !out || !out_t) VITK_FAIL(VITK_E_ARG, "pack_w_nt_many: null table");
    hipStream_t st = (hipStream_t)stream;
    PackMany a;
    a.count = 0; a.blk0[0] = 0;
    for (int i = 0; i < PM_MAX / 32; ++i) a.tr[i] = 0u;
    auto flush = [&]() -> int {
        if (a.count == 0) return 0;
        hipLaunchKernelGGL(pack_w_many_kernel, dim3((unsigned)a.blk0[a.count]), dim3(256), 0, st, a);
        VITK_CHECK_LAUNCH("pack_w_nt_many");
        a.count = 0; a.blk0[0] = 0;
        for (int i = 0; i < PM_MAX / 32; ++i) a.tr[i] = 0u;
        return 0;
    };
    for (int64_t t = 0; t < count; ++t) {
        if (!W[t] || (!out[t] && !out_t[t])) VITK_FAIL(VITK_E_ARG, "pack_w_nt_many: null pointer in row %lld", (long long)t);
        if (N[t] <= 0 || K[t] <= 0 || ldw[t] < K[t] || (ldw[t] & 7) || N[t] > (1 << 24) || K[t] > (1 << 24))
            VITK_FAIL(VITK_E_SHAPE, "pack_w_nt_many: bad shape N=%lld K=%lld ldw=%lld", (long long)N[t], (long long)K[t], (long long)ldw[t]);
        if ((out[t] && (K[t] & 31)) || (out_t[t] && (N[t] & 31))) VITK_FAIL(VITK_E_SHAPE, "pack_w_nt_many: the reduction dimension must be a multiple of 32");
        if (!aligned16(W[t]) || (out[t] && !aligned16(out[t])) || (out_t[t] && !aligned16(out_t[t]))) VITK_FAIL(VITK_E_ALIGN, "pack_w_nt_many: 16-byte aligned pointers required");
        for (int tr = 0; tr < 2; ++tr) {
            void* o = tr ? out_t[t] : out[t];
            if (!o) continue;
            const long long chunks = (tr ? vitk_pack_w_nt_bytes(K[t], N[t]) : vitk_pack_w_nt_bytes(N[t], K[t])) / 16;
            const long long blocks = (chunks + 255) / 256;
            if (a.count == PM_MAX || a.blk0[a.count] + blocks > 0x3fffffff) { if (int rc = flush()) return rc; }
            const int j = a.count;
            a.W[j] = (const __bf16*)W[t]; a.out[j] = (char*)o; a.ldw[j] = (int)ldw[t]; a.N[j] = (int)N[t]; a.K[j] = (int)K[t];
            if (tr) a.tr[j >> 5] |= 1u << (j & 31);
            a.blk0[j + 1] = a.blk0[j] + (int)blocks;
            ++a.count;
        }
    }
    return flush();
}
