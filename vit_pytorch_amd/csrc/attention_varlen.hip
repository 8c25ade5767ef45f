// attention_varlen.hip -- variable-length (packed) flash attention forward / backward for gfx950, 16-bit, dim_head 32 .. 96.
//
// The NaViT path (na_vit.py:115-169, 255-402) and every fixed-length shape the whole-head kernels of attention.hip /
// attention_pipe.hip do not take (ViT-H/14: dim_head 80, N = 577; vit.py:55-63 with vit.py:86's free dim_head).
// The reference pads every pack to (b, n) and feeds F.scaled_dot_product_attention a dense boolean (b, 1, n, n) mask
// "same image AND key not padding" (na_vit.py:335-337).  Here tokens of all images of all packs live UNPADDED in one
// (T, H*d) matrix and attention is computed per SEGMENT (= image): query rows [cu_q[s], cu_q[s+1]) against key rows
// [cu_k[s], cu_k[s+1]).  No mask is ever built or read: the block-diagonal structure is the launch geometry.  The
// attention-pool step (one learned query per image against that image's tokens, na_vit.py:371-387) is the same kernel with
// cu_q = 0,1,2,...  A fixed-length batch is B segments of N rows (segments.uniform_segments).
//
// One workgroup = (segment, 128-row block, head).  Fragment algebra and softmax are those of attention.hip (S^T = K Q^T, P^T
// feeding P.V straight from the accumulators, lazy reference maximum, MFMA row sums; backward as a query-block dQ kernel and a
// key-block dK/dV kernel that recompute P from the saved log-sum-exp).  The OTHER operand pair (K|V for the forward and dQ,
// Q|dO for dK/dV) streams through LDS in 64-row chunks.
//
// Round 4 (the kernels of rounds 2-3 staged 128-row chunks global -> registers -> ds_write behind two barriers per chunk, every
// fragment read fed one MFMA, and the dK/dV kernel -- held to 128 registers for four waves per SIMD -- kept 42 (d = 80) / 47
// (d = 96) registers in scratch: config 5's attention ran at 250-450 TF/s):
//  * chunks arrive by LDS-DMA (global_load_lds, 16 bytes per lane, no VGPR round trip, no ds_write issue) into TWO buffers:
//    chunk c + 1 is in flight while chunk c is multiplied, ONE barrier per chunk, no staging registers.  A wave instruction
//    writes 1 KiB of the LDS image lane-linearly, so the image's row padding is filled too: lanes that fall into a row's pad
//    re-read the start of that row.  The columns d .. 32 NKS of the contraction are ZERO IN THE REGISTER OPERAND, so what the
//    LDS operand holds there (pad, or the start of the next row) never reaches a result as long as it is finite -- it is a
//    copy of the tensor's own data.  Rows past the end of the segment re-read its last row; they are masked by index as before.
//  * geometry (R, NW): a workgroup is NW waves, each carrying R 16-row tiles of the block (16 R NW = 128).  (2, 4), the
//    default: every K / V / Q / dO fragment feeds TWO MFMAs (half the LDS reads and half the ds_read issue per flop), four
//    independent MFMA chains per step, four-wave workgroups (two or three to a CU, each at its own phase).  (1, 8): the
//    geometry of rounds 2-3 on the new staging (VITK_ATTN_VL=1, A/B runs).
//  * the LDS row pitch is the smallest 32 x odd bytes that holds a row (96 / 96 / 160 / 160 / 224 for d = 32 / 48 / 64 / 80 /
//    96): consecutive rows then start 8 banks x odd apart, which makes BOTH fragment shapes conflict-free -- ds_read_b128 row
//    fragments (16 rows x one 16-byte slot) and ds_read_b64_tr_b16 transpose fragments (8 rows x 32 bytes per 32-lane half).
//    The 208-byte rows used before for d = 80 / 96 were conflict-free for the row fragments only (2-way on the transposing
//    reads: rows r and r + 5 overlap), and at 160 bytes three workgroups of d = 80 fit a CU's LDS instead of two.
#include "common.h"
#include "attention_frag.h"
#include <stdlib.h>

namespace {

#ifndef VL_EARLY_TR
#define VL_EARLY_TR 0        // 1: the transposing reads of a step are issued at its top (in flight under the S / dP MFMAs and the
#endif                       //    softmax arithmetic), 0: right in front of their MFMAs.  [measured, profiles/r04f_vl_bench_early.log]
                             //    equal within 2 % either way; 0 needs 8-16 registers less
constexpr int VL_BLK = 128;            // rows per block (host tables: segments.Segments.BLOCK)
constexpr int VL_CH = 64;              // rows per LDS chunk

struct HND { __bf16* p; long long s_h, s_n; };   // element (n, h, d) at p + n*s_n + h*s_h + d

// Head-dimension traits.  DH = 80 (ViT-H/14, vit.py dim_head=80): the contraction over d is padded to 96 = 3 K-steps (the
// register operand is zero there), the output has 5 blocks of 16 columns.
template <int DH> struct HD {
    static constexpr int NKS = (DH + 31) / 32;      // MFMA K-steps over d
    static constexpr int NFD = DH / 16;             // 16-column output blocks
    // bytes per LDS row: the smallest 32 x odd >= the row's data (96 / 96 / 160 / 160 / 224 for d = 32 / 48 / 64 / 80 / 96).  Row
    // fragments read 64 NKS bytes of a row: for d = 48 / 80 that is 32 bytes into the NEXT row (finite data against zero columns
    // of the register operand); past the last tile lies a zeroed 32-byte slack (SLACK).  The last row of buffer 0's second tile
    // reads into the START OF BUFFER 1, which a segment of one chunk (<= 64 rows) never stages: the kernels zero those 32 bytes
    // too before their first barrier (found by tests/test_fuzz_gpu.py in IEEE half: stale LDS bits there are Inf / NaN patterns one
    // time in 32, and 0 x NaN reached dK / dV / dQ)
    static constexpr int LD = ((DH * 2 + 31) / 32) % 2 ? (DH * 2 + 31) / 32 * 32 : (DH * 2 + 31) / 32 * 32 + 32;
    static_assert(LD >= DH * 2 && LD + 32 >= NKS * 64 && (LD / 32) % 2 == 1, "row pitch");
    static constexpr int SLACK = NKS * 64 > LD ? 32 : 0;
    static constexpr int TILE = VL_CH * LD;         // one chunk of one tensor
    static constexpr int NI = TILE / 1024;          // wave-wide DMA instructions per chunk and tensor
    static_assert(NI * 1024 == TILE, "a chunk is a whole number of 1 KiB DMA pieces");
};

// chunk rows [0, rows) of a (n, h, d) tensor (base = its first row, head applied) -> LDS tile; rows >= `rows` re-read row rows - 1.
// Piece i = wave + j NW of the tile is 1 KiB of the image; this lane's 16 bytes of it are (row, column) = (off / LD, off % LD) with
// off = 1024 i + 16 lane -- recomputed per chunk (a handful of VALU) rather than kept in registers; pad bytes re-read the start of
// their row.
template <int DH, int NW>
__device__ __forceinline__ void dma_chunk(const __bf16* base, long long s_n, int rows, char* tile, int wave, int lane) {
    constexpr int LD = HD<DH>::LD, NJ = (HD<DH>::NI + NW - 1) / NW;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int i = wave + j * NW;
        if (i < HD<DH>::NI) {          // wave-uniform
            const int off = i * 1024 + lane * 16;
            int r = off / LD;
            int c = off - r * LD;
            if (c >= DH * 2) c -= DH * 2;
            r = r < rows ? r : rows - 1;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(base + (long long)r * s_n + (c >> 1)),
                                             (void __attribute__((address_space(3)))*)(tile + i * 1024), 16, 0, 0);
        }
    }
}
// `rows` floats -> LDS (one wave, one instruction: VL_CH = 64 lanes x 4 bytes); lanes >= rows re-read element rows - 1
__device__ __forceinline__ void dma_f32_64(const float* base, int rows, char* dst, int lane) {
    const int i = lane < rows ? lane : rows - 1;
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(base + i),
                                     (void __attribute__((address_space(3)))*)dst, 4, 0, 0);
}
// every DMA of this wave has landed, every wave's fragment reads of the buffer about to be refilled are done; the clobbers tell
// the compiler that LDS changed behind its back (the compute code never writes LDS)
#define VL_SYNC() do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); \
    __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); } while (0)

// 16 rows x 32 columns as an MFMA A operand: lane (i, g) holds tile[row0 + i][ks*32 + 8g .. +7]
template <int NKS, int LD>
__device__ __forceinline__ void read_rows(bf16x8 (&f)[NKS], const char* tile, int row0, int fi, int fg) {
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) f[ks] = *reinterpret_cast<const bf16x8*>(tile + (row0 + fi) * LD + (ks * 32 + 8 * fg) * 2);
}
// Transposed operands: fragment fd of a 32-row step, lane (i, g) holds tile[row0 + {4g..4g+3, 16+4g..16+4g+3}][16 fd + i] (two
// ds_read_b64_tr_b16).  Issued as ASM: in front of every ds_read_tr BUILTIN hipcc puts s_waitcnt vmcnt(0) while an LDS-DMA of the
// same wave is in flight (it cannot tell the buffers apart) -- i.e. every step of chunk c would wait for chunk c + 1 and nothing
// would overlap.  The compiler neither sees these reads as LDS accesses nor counts them: all fragments of a block are issued, then
// ONE s_waitcnt lgkmcnt(0) that names every destination as "+v" (so no consumer can be scheduled above it), then the MFMAs.
template <int OFF> __device__ __forceinline__ s16x4 tr_asm(unsigned lds_addr) {
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(OFF) : "memory");
    return v;
}
template <int LD> __device__ __forceinline__ unsigned tr_lane_off(int fi, int fg) { return (unsigned)((4 * fg + (fi >> 2)) * LD + (fi & 3) * 8); }
__device__ __forceinline__ unsigned lds_addr_of(const char* p) { return (unsigned)(uintptr_t)((const __attribute__((address_space(3))) char*)p); }
template <int NFD, int LD>
__device__ __forceinline__ void tr_read_all(s16x4 (&lo)[NFD], s16x4 (&hi)[NFD], unsigned addr) {
    static_assert(NFD >= 2 && NFD <= 6, "dim_head 32 .. 96");
    lo[0] = tr_asm<0>(addr); hi[0] = tr_asm<16 * LD>(addr);
    lo[1] = tr_asm<32>(addr); hi[1] = tr_asm<32 + 16 * LD>(addr);
    if constexpr (NFD > 2) { lo[2] = tr_asm<64>(addr); hi[2] = tr_asm<64 + 16 * LD>(addr); }
    if constexpr (NFD > 3) { lo[3] = tr_asm<96>(addr); hi[3] = tr_asm<96 + 16 * LD>(addr); }
    if constexpr (NFD > 4) { lo[4] = tr_asm<128>(addr); hi[4] = tr_asm<128 + 16 * LD>(addr); }
    if constexpr (NFD > 5) { lo[5] = tr_asm<160>(addr); hi[5] = tr_asm<160 + 16 * LD>(addr); }
}
#define VL_LG0 "s_waitcnt lgkmcnt(0)"
template <int NFD>
__device__ __forceinline__ void tr_wait(s16x4 (&a)[NFD], s16x4 (&b)[NFD]) {
    if constexpr (NFD == 2) asm volatile(VL_LG0 : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]) :: "memory");
    else if constexpr (NFD == 3) asm volatile(VL_LG0 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]) :: "memory");
    else if constexpr (NFD == 4) asm volatile(VL_LG0 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) :: "memory");
    else if constexpr (NFD == 5) asm volatile(VL_LG0 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]) :: "memory");
    else asm volatile(VL_LG0 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]) :: "memory");
}
__device__ __forceinline__ bf16x8 tr_join(s16x4 lo, s16x4 hi) {
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
// one row of a (n, h, d) tensor as NKS register fragments: lane (i, g) holds row[ks*32 + 8g .. +7], ZERO beyond DH (this is what
// keeps the padded columns of the LDS operand out of every contraction over d)
template <int DH, int NKS>
__device__ __forceinline__ void load_row_frags(bf16x8 (&f)[NKS], const __bf16* rowp, int fg) {
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) f[ks] = (ks * 32 + 8 * fg < DH) ? *reinterpret_cast<const bf16x8*>(rowp + ks * 32 + 8 * fg) : zero8;
}
template <int NKS>
__device__ __forceinline__ f32x4 mfma_over_d(const bf16x8 (&a)[NKS], const bf16x8 (&b)[NKS]) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) acc = MFMA(a[ks], b[ks], acc);
    return acc;
}

// halves of a B operand, rounded as soon as they exist (2 registers instead of 4 while the other half is being formed)
__device__ __forceinline__ bf16x4 pack4(f32x4 a) { return bf16x4{(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3]}; }
__device__ __forceinline__ bf16x8 cat8(bf16x4 a, bf16x4 b) { return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }

// this wave's tile r of the block: row inside the segment (may lie past its end: results of such rows are not stored) and the
// global (packed) row clamped into the segment (loads of rows past the end re-read its last row).  Recomputed where needed (two or
// three VALU) instead of being carried through the main loop in registers.
struct Tiles {
    int t0, fi, seg0, n;
    __device__ __forceinline__ int row(int r) const { return t0 + r * 16 + fi; }
    __device__ __forceinline__ int grow(int r) const { const int x = row(r); return seg0 + (x < n ? x : (n > 0 ? n - 1 : 0)); }
};

// Which (block, head) a workgroup takes.  The blocks of one (segment, head) stream the SAME K|V (or Q|dO) rows, and the hardware
// deals consecutive workgroup ids round-robin over the 8 XCDs, each with its own L2: with (block, head) = plain grid coordinates the
// five blocks of a ViT-H/14 head ran on five XCDs and every one of them pulled the head's 185 KB through the fabric -- [measured,
// profiles/r04f_vl_ablation.log] the kernels were bound by that staging (forward 248 us at batch 64, of which LDS-DMA + barriers
// alone 236 us, arithmetic alone 156 us).  Here XCD x = id % 8 walks the contiguous range [x per, (x + 1) per) of an ORDER in which
// the blocks that share rows are neighbours: they run back to back on one XCD and all but the first hit its L2.
constexpr int VL_GROUP = 16;
// order 2 (the default): the heads in 8 chunks, inside a chunk groups of VL_GROUP blocks x the chunk's heads -- an XCD takes H / 8
// heads of ALL segments (balanced whatever the segment lengths) and works on them side by side (neighbouring heads share 128-byte
// lines).  Order 0: head-major over all blocks (an XCD takes its heads one after the other); order 1: groups of VL_GROUP blocks x all
// heads (an XCD takes a range of SEGMENTS: whole token rows, but their lengths decide the balance).  [measured,
// profiles/r04f_vl_orders.log] ViT-H/14 shape, batch 256, forward / backward us: order 0: 958 / 2,462, 1: 905 / 2,373, 2: 893 / 2,339;
// NaViT mix of 48 images: 116 / 364, 160 / 439, 116 / 367.  VITK_VL_ORDER overrides.
__device__ __forceinline__ bool vl_block_of(int nblk, int nheads, int order, int& blk, int& h) {
    const int per = gridDim.x >> 3;
    int v = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (v >= nblk * nheads) return false;
    if (order == 0) { h = v / nblk; blk = v - h * nblk; return true; }
    int h0 = 0, hc = nheads;
    if (order == 2) {
        const int q = v / nblk;             // chunk c holds the heads [c H / 8, (c + 1) H / 8)
        int c = 0;
        while (((c + 1) * nheads) / 8 <= q) ++c;
        h0 = (c * nheads) / 8; hc = ((c + 1) * nheads) / 8 - h0;
        v -= h0 * nblk;
    }
    const int g = v / (hc * VL_GROUP);
    const int rem = v - g * (hc * VL_GROUP);
    const int left = nblk - g * VL_GROUP;
    const int gl = left < VL_GROUP ? left : VL_GROUP;       // blocks in this group (the last one may be short)
    const int hl = rem / gl;
    h = h0 + hl;
    blk = g * VL_GROUP + (rem - hl * gl);
    return true;
}

// ------------------------------------------------------------------------------------------------------------------------
// forward: O = softmax(scale Q K^T) V, row log-sum-exp saved for the backward
// ------------------------------------------------------------------------------------------------------------------------
template <int DH, int R, int NW, bool DROP>
__global__ __launch_bounds__(64 * NW, (NW == 8 && DH <= 80 && !DROP && !VL_EARLY_TR) ? 6 : 2) void attn_varlen_fwd_kernel(
    HND q, HND k, HND v, HND o, float* __restrict__ lse, const int* __restrict__ cu_q, const int* __restrict__ cu_k,
    const int* __restrict__ blk_seg, const int* __restrict__ blk_r0, int tq_total, float scale_log2e, unsigned drop_t,
    unsigned drop_seed, float inv_keep, int nblk, int nheads, int dbg) {
    using T = HD<DH>;
    constexpr int NKS = T::NKS, NFD = T::NFD, LD = T::LD, TILE = T::TILE;
    static_assert(16 * R * NW == VL_BLK, "a block is 128 rows");
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE + T::SLACK];      // buffer b: K at 2 b TILE, V at (2 b + 1) TILE
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, fg = lane >> 4;
    const unsigned troff = tr_lane_off<LD>(fi, fg);
    // (block, head) of this workgroup: see vl_block_of
    int blk, h;
    if (!vl_block_of(nblk, nheads, dbg >> 8, blk, h)) return;
    const int seg = blk_seg[blk];
    const int qs = cu_q[seg], nq = cu_q[seg + 1] - qs;
    const int ks0 = cu_k[seg], nk = cu_k[seg + 1] - ks0;
    const int t0 = blk_r0[blk] + wave * (16 * R);       // first row of this wave's first tile
    const bool wave_active = t0 < nq;
    const __bf16* kbase = k.p + (long long)ks0 * k.s_n + h * k.s_h;
    const __bf16* vbase = v.p + (long long)ks0 * v.s_n + h * v.s_h;
    if (tid < T::SLACK / 4) {       // published by the first VL_SYNC; buffer 1 is first staged behind that barrier
        reinterpret_cast<unsigned*>(smem + 4 * TILE)[tid] = 0u;
        reinterpret_cast<unsigned*>(smem + 2 * TILE)[tid] = 0u;        // a ONE-chunk segment never stages buffer 1: see HD::SLACK
    }
    auto stage = [&](int c0, int buf) {
        const int rows = nk - c0 < VL_CH ? nk - c0 : VL_CH;
        dma_chunk<DH, NW>(kbase + (long long)c0 * k.s_n, k.s_n, rows, smem + 2 * buf * TILE, wave, lane);
        dma_chunk<DH, NW>(vbase + (long long)c0 * v.s_n, v.s_n, rows, smem + (2 * buf + 1) * TILE, wave, lane);
    };
    if (nk > 0) stage(0, 0);
    const Tiles tl{t0, fi, qs, nq};
    bf16x8 qf[R][NKS];
#pragma unroll
    for (int r = 0; r < R; ++r) load_row_frags<DH, NKS>(qf[r], q.p + (long long)tl.grow(r) * q.s_n + h * q.s_h, fg);
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    bf16x8 ones = {(__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f};
    asm volatile("" : "+v"(ones));                    // four live registers (hipcc otherwise rebuilds the operand in front of every row-sum MFMA)
    const float c = scale_log2e;                      // > 0 (host check)
    const float inv_c = 1.0f / c;
    float mref[R], thr[R];                            // lazy reference maximum, row sums by MFMA: see attn_fwd_kernel (attention.hip); thr = (mref + 8) / c
    f32x4 acc[R][NFD], accl[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        mref[r] = -INFINITY; thr[r] = -INFINITY; accl[r] = z4;
#pragma unroll
        for (int fd = 0; fd < NFD; ++fd) acc[r][fd] = z4;
    }
    int buf = 0;
    for (int c0 = 0; c0 < nk; c0 += VL_CH, buf ^= 1) {
        const int rows = nk - c0 < VL_CH ? nk - c0 : VL_CH;
        const int rows_pad = ((rows + 31) >> 5) << 5;
        if (!(dbg & 4) || c0 == 0) VL_SYNC();         // chunk c0 has landed; the other buffer has been read by everyone
        if (c0 + VL_CH < nk && !(dbg & 1)) stage(c0 + VL_CH, buf ^ 1);
        if (!wave_active || (dbg & 2)) continue;
        const char* Ks = smem + 2 * buf * TILE;
        const char* Vs = Ks + TILE;
        for (int s = 0; s < (rows_pad >> 5); ++s) {
            bf16x8 kf[2][NKS];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) read_rows<NKS, LD>(kf[hh], Ks, s * 32 + hh * 16, fi, fg);
            s16x4 lo[NFD], hi[NFD];
            if constexpr (VL_EARLY_TR) tr_read_all<NFD, LD>(lo, hi, lds_addr_of(Vs) + troff + s * (32 * LD));
            // The R tiles go through the step phase by phase (scores, decision, exponentials) so that each phase is ONE basic block in which
            // hipcc interleaves the R independent chains; the rare raise of the reference maximum is decided for all tiles at once.
            bf16x8 pb[R];
            f32x4 st[R][2];
            float mloc[R];
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) st[r][hh] = mfma_over_d<NKS>(kf[hh], qf[r]);
            if (s * 32 + 32 > rows) {                 // padding keys: only in the last step of the last chunk
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (s * 32 + hh * 16 + 4 * fg + e >= rows) st[r][hh][e] = -INFINITY;
            }
            bool raise = false;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float m = fmaxf(fmaxf(st[r][0][0], st[r][0][1]), st[r][0][2]);
                m = fmaxf(fmaxf(m, st[r][0][3]), st[r][1][0]);
                m = fmaxf(fmaxf(m, st[r][1][1]), st[r][1][2]);
                mloc[r] = fmaxf(m, st[r][1][3]);
                raise = raise || (mloc[r] > thr[r]);
            }
            if (__builtin_amdgcn_ballot_w64(raise) != 0) {
#pragma unroll
                for (int r = 0; r < R; ++r) {         // a tile whose scores did not call for it is rescaled by a harmless alpha <= 1 too
                    const float m_new = fmaxf(mref[r], groups_max(mloc[r]) * c);
                    const float alpha = __builtin_amdgcn_exp2f(mref[r] - m_new);
#pragma unroll
                    for (int fd = 0; fd < NFD; ++fd) acc[r][fd] *= alpha;
                    accl[r] *= alpha;
                    mref[r] = m_new;
                    thr[r] = (m_new + 8.0f) * inv_c;
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float nm = -mref[r];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                    for (int e = 0; e < 4; ++e) st[r][hh][e] = __builtin_amdgcn_exp2f(fmaf(st[r][hh][e], c, nm));
                pb[r] = pack8(st[r][0], st[r][1]);
                accl[r] = MFMA(ones, pb[r], accl[r]);          // softmax denominators: of the UNDROPPED probabilities
                if constexpr (DROP) {     // attention dropout (na_vit.py:163 dropout_p): row = (head, global query row), column = key inside the image
                    const unsigned hrow = drop_row((unsigned)(h * tq_total + tl.grow(r)), drop_seed);
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (!drop_keep(hrow, (unsigned)(c0 + s * 32 + hh * 16 + 4 * fg + e), drop_t)) st[r][hh][e] = 0.f;
                    pb[r] = pack8(st[r][0], st[r][1]);
                }
            }
            {
                if constexpr (!VL_EARLY_TR) tr_read_all<NFD, LD>(lo, hi, lds_addr_of(Vs) + troff + s * (32 * LD));
                tr_wait<NFD>(lo, hi);
#pragma unroll
                for (int fd = 0; fd < NFD; ++fd) {
                    const bf16x8 vf = tr_join(lo[fd], hi[fd]);
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[r][fd] = MFMA(vf, pb[r], acc[r][fd]);
                }
            }
        }
    }
    if (!wave_active) return;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const bool ok = tl.row(r) < nq;
        const float ls = accl[r][0];        // every row of 1^T P^T is the same sum
        const float inv = inv_keep / ls;    // kept entries are scaled by 1 / (1 - p)
        store_rows16<NFD>(o.p + (long long)tl.grow(r) * o.s_n + h * o.s_h, acc[r], inv, fg, ok);
        if (ok && fg == 0) lse[(long long)h * tq_total + tl.grow(r)] = (mref[r] + log2f(ls)) * LN2;
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// backward, query-block outer: dQ = scale dS K with dS = P (dP - delta); writes delta = rowsum(dO O) for the dK/dV kernel
// ------------------------------------------------------------------------------------------------------------------------
template <int DH, int R, int NW, bool DROP>
__global__ __launch_bounds__(64 * NW, (NW == 4 && DH <= 80 && !VL_EARLY_TR) ? 3 : 2) void attn_varlen_bwd_dq_kernel(
    HND q, HND k, HND v, HND o, HND dout, const float* __restrict__ lse, float* __restrict__ delta, HND dq,
    const int* __restrict__ cu_q, const int* __restrict__ cu_k, const int* __restrict__ blk_seg,
    const int* __restrict__ blk_r0, int tq_total, float scale, unsigned drop_t, unsigned drop_seed, float inv_keep, int nblk, int nheads, int dbg) {
    using T = HD<DH>;
    constexpr int NKS = T::NKS, NFD = T::NFD, LD = T::LD, TILE = T::TILE;
    static_assert(16 * R * NW == VL_BLK, "a block is 128 rows");
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE + T::SLACK];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, fg = lane >> 4;
    const unsigned troff = tr_lane_off<LD>(fi, fg);
    // (block, head) of this workgroup: see vl_block_of
    int blk, h;
    if (!vl_block_of(nblk, nheads, dbg >> 8, blk, h)) return;
    const int seg = blk_seg[blk];
    const int qs = cu_q[seg], nq = cu_q[seg + 1] - qs;
    const int ks0 = cu_k[seg], nk = cu_k[seg + 1] - ks0;
    const int t0 = blk_r0[blk] + wave * (16 * R);
    const bool wave_active = t0 < nq;
    const __bf16* kbase = k.p + (long long)ks0 * k.s_n + h * k.s_h;
    const __bf16* vbase = v.p + (long long)ks0 * v.s_n + h * v.s_h;
    if (tid < T::SLACK / 4) {       // published by the first VL_SYNC; buffer 1 is first staged behind that barrier
        reinterpret_cast<unsigned*>(smem + 4 * TILE)[tid] = 0u;
        reinterpret_cast<unsigned*>(smem + 2 * TILE)[tid] = 0u;        // a ONE-chunk segment never stages buffer 1: see HD::SLACK
    }
    auto stage = [&](int c0, int buf) {
        const int rows = nk - c0 < VL_CH ? nk - c0 : VL_CH;
        dma_chunk<DH, NW>(kbase + (long long)c0 * k.s_n, k.s_n, rows, smem + 2 * buf * TILE, wave, lane);
        dma_chunk<DH, NW>(vbase + (long long)c0 * v.s_n, v.s_n, rows, smem + (2 * buf + 1) * TILE, wave, lane);
    };
    if (nk > 0) stage(0, 0);
    const Tiles tl{t0, fi, qs, nq};
    bf16x8 qf[R][NKS], df[R][NKS];
    float dl[R], nl2[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        bf16x8 of[NKS];
        load_row_frags<DH, NKS>(qf[r], q.p + (long long)tl.grow(r) * q.s_n + h * q.s_h, fg);
        load_row_frags<DH, NKS>(df[r], dout.p + (long long)tl.grow(r) * dout.s_n + h * dout.s_h, fg);
        load_row_frags<DH, NKS>(of, o.p + (long long)tl.grow(r) * o.s_n + h * o.s_h, fg);
        float d = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) d += dot8(df[r][ks], of[ks]);
        dl[r] = groups_sum(d);
        if (wave_active && tl.row(r) < nq && fg == 0) delta[(long long)h * tq_total + tl.grow(r)] = dl[r];
        nl2[r] = -lse[(long long)h * tq_total + tl.grow(r)] * LOG2E;
    }
    const float scale_log2e = scale * LOG2E;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[R][NFD];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int fd = 0; fd < NFD; ++fd) acc[r][fd] = z4;
    int buf = 0;
    for (int c0 = 0; c0 < nk; c0 += VL_CH, buf ^= 1) {
        const int rows = nk - c0 < VL_CH ? nk - c0 : VL_CH;
        const int rows_pad = ((rows + 31) >> 5) << 5;
        if (!(dbg & 4) || c0 == 0) VL_SYNC();
        if (c0 + VL_CH < nk && !(dbg & 1)) stage(c0 + VL_CH, buf ^ 1);
        if (!wave_active || (dbg & 2)) continue;
        const char* Ks = smem + 2 * buf * TILE;
        const char* Vs = Ks + TILE;
        for (int s = 0; s < (rows_pad >> 5); ++s) {
            bf16x8 dsb[R];
            s16x4 lo[NFD], hi[NFD];
            if constexpr (VL_EARLY_TR) tr_read_all<NFD, LD>(lo, hi, lds_addr_of(Ks) + troff + s * (32 * LD));
            {
                bf16x4 dsh[R][2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int row0 = s * 32 + hh * 16;
                    bf16x8 kfr[NKS], vfr[NKS];
                    read_rows<NKS, LD>(kfr, Ks, row0, fi, fg);
                    read_rows<NKS, LD>(vfr, Vs, row0, fi, fg);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const f32x4 st = mfma_over_d<NKS>(kfr, qf[r]);
                        f32x4 dp = mfma_over_d<NKS>(vfr, df[r]);
                        if constexpr (DROP) {
                            const unsigned hrow = drop_row((unsigned)(h * tq_total + tl.grow(r)), drop_seed);
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                dp[e] = drop_keep(hrow, (unsigned)(c0 + row0 + 4 * fg + e), drop_t) ? dp[e] * inv_keep : 0.f;
                        }
                        f32x4 ds;
#pragma unroll
                        for (int e = 0; e < 4; ++e)          // `scale` of dS is applied once, to dQ
                            ds[e] = __builtin_amdgcn_exp2f(fmaf(st[e], scale_log2e, nl2[r])) * (dp[e] - dl[r]);
                        if (row0 + 16 > rows) {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (row0 + 4 * fg + e >= rows) ds[e] = 0.f;
                        }
                        dsh[r][hh] = pack4(ds);
                    }
                }
#pragma unroll
                for (int r = 0; r < R; ++r) dsb[r] = cat8(dsh[r][0], dsh[r][1]);
            }
            {
                if constexpr (!VL_EARLY_TR) tr_read_all<NFD, LD>(lo, hi, lds_addr_of(Ks) + troff + s * (32 * LD));
                tr_wait<NFD>(lo, hi);
#pragma unroll
                for (int fd = 0; fd < NFD; ++fd) {
                    const bf16x8 kt = tr_join(lo[fd], hi[fd]);
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[r][fd] = MFMA(kt, dsb[r], acc[r][fd]);
                }
            }
        }
    }
    if (!wave_active) return;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        // (8-byte stores here: with store_rows16 the d = 80 dropout instance of this kernel spills inside its key loop)
        if (tl.row(r) < nq) {
            __bf16* dqp = dq.p + (long long)tl.grow(r) * dq.s_n + h * dq.s_h + 4 * fg;
#pragma unroll
            for (int fd = 0; fd < NFD; ++fd) store4<__bf16>(dqp + fd * 16, acc[r][fd] * scale);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// backward, key-block outer: dV = P^T dO, dK = scale dS^T Q (reads the delta the dQ kernel wrote)
// ------------------------------------------------------------------------------------------------------------------------
template <int DH, int R, int NW, bool DROP>
__global__ __launch_bounds__(64 * NW, 2) void attn_varlen_bwd_dkv_kernel(
    HND q, HND k, HND v, HND dout, const float* __restrict__ lse, const float* __restrict__ delta, HND dk, HND dv,
    const int* __restrict__ cu_q, const int* __restrict__ cu_k, const int* __restrict__ blk_seg,
    const int* __restrict__ blk_r0, int tq_total, float scale, unsigned drop_t, unsigned drop_seed, float inv_keep, int nblk, int nheads, int dbg) {
    using T = HD<DH>;
    constexpr int NKS = T::NKS, NFD = T::NFD, LD = T::LD, TILE = T::TILE;
    static_assert(16 * R * NW == VL_BLK && NW >= 2, "a block is 128 rows; waves 0 and 1 stage lse / delta");
    // buffer b: Q at 2 b TILE, dO at (2 b + 1) TILE; behind the tiles and their slack: lse of the chunk's rows at 512 b, delta 256 bytes further
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE + T::SLACK + 1024];
    char* const stats = smem + 4 * TILE + T::SLACK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, fg = lane >> 4;
    const unsigned troff = tr_lane_off<LD>(fi, fg);
    // (block, head) of this workgroup: see vl_block_of
    int blk, h;
    if (!vl_block_of(nblk, nheads, dbg >> 8, blk, h)) return;
    const int seg = blk_seg[blk];
    const int qs = cu_q[seg], nq = cu_q[seg + 1] - qs;
    const int ks0 = cu_k[seg], nk = cu_k[seg + 1] - ks0;
    const int t0 = blk_r0[blk] + wave * (16 * R);       // first key row (inside the segment) of this wave's first tile
    const bool wave_active = t0 < nk;
    const __bf16* qbase = q.p + (long long)qs * q.s_n + h * q.s_h;
    const __bf16* dbase = dout.p + (long long)qs * dout.s_n + h * dout.s_h;
    const float* lbase = lse + (long long)h * tq_total + qs;
    const float* dlbase = delta + (long long)h * tq_total + qs;
    if (tid < T::SLACK / 4) {       // published by the first VL_SYNC; buffer 1 is first staged behind that barrier
        reinterpret_cast<unsigned*>(smem + 4 * TILE)[tid] = 0u;
        reinterpret_cast<unsigned*>(smem + 2 * TILE)[tid] = 0u;        // a ONE-chunk segment never stages buffer 1: see HD::SLACK
    }
    auto stage = [&](int c0, int buf) {
        const int rows = nq - c0 < VL_CH ? nq - c0 : VL_CH;
        dma_chunk<DH, NW>(qbase + (long long)c0 * q.s_n, q.s_n, rows, smem + 2 * buf * TILE, wave, lane);
        dma_chunk<DH, NW>(dbase + (long long)c0 * dout.s_n, dout.s_n, rows, smem + (2 * buf + 1) * TILE, wave, lane);
        if (wave == 0) dma_f32_64(lbase + c0, rows, stats + 512 * buf, lane);
        if (wave == 1) dma_f32_64(dlbase + c0, rows, stats + 512 * buf + 256, lane);
    };
    if (nq > 0) stage(0, 0);
    const Tiles tl{t0, fi, ks0, nk};
    bf16x8 kf[R][NKS], vf[R][NKS];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        load_row_frags<DH, NKS>(kf[r], k.p + (long long)tl.grow(r) * k.s_n + h * k.s_h, fg);
        load_row_frags<DH, NKS>(vf[r], v.p + (long long)tl.grow(r) * v.s_n + h * v.s_h, fg);
    }
    const float scale_log2e = scale * LOG2E;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 accK[R][NFD], accV[R][NFD];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int fd = 0; fd < NFD; ++fd) { accK[r][fd] = z4; accV[r][fd] = z4; }
    int buf = 0;
    for (int c0 = 0; c0 < nq; c0 += VL_CH, buf ^= 1) {
        const int rows = nq - c0 < VL_CH ? nq - c0 : VL_CH;
        const int rows_pad = ((rows + 31) >> 5) << 5;
        if (!(dbg & 4) || c0 == 0) VL_SYNC();
        if (c0 + VL_CH < nq && !(dbg & 1)) stage(c0 + VL_CH, buf ^ 1);
        if (!wave_active || (dbg & 2)) continue;
        const char* Qs = smem + 2 * buf * TILE;
        const char* Ds = Qs + TILE;
        const float* lse_s = reinterpret_cast<const float*>(stats + 512 * buf);
        const float* del_s = lse_s + 64;
        for (int s = 0; s < (rows_pad >> 5); ++s) {
            bf16x8 pb[R], dsb[R];
            {
                bf16x4 ph[R][2], dsh[R][2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int row0 = s * 32 + hh * 16;
                    bf16x8 qfr[NKS], dfr[NKS];
                    read_rows<NKS, LD>(qfr, Qs, row0, fi, fg);
                    read_rows<NKS, LD>(dfr, Ds, row0, fi, fg);
                    const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + row0 + 4 * fg) * (-LOG2E);
                    const f32x4 d4 = *reinterpret_cast<const f32x4*>(del_s + row0 + 4 * fg);
                    unsigned hq[4] = {0u, 0u, 0u, 0u};         // dropout row hashes of this lane's four query rows
                    if constexpr (DROP) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) hq[e] = drop_row((unsigned)(h * tq_total + qs + c0 + row0 + 4 * fg + e), drop_seed);
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const f32x4 st = mfma_over_d<NKS>(qfr, kf[r]);
                        const f32x4 dp = mfma_over_d<NKS>(dfr, vf[r]);
                        f32x4 p, ds;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            p[e] = __builtin_amdgcn_exp2f(fmaf(st[e], scale_log2e, l4[e]));
                            if constexpr (DROP) {
                                const float km = drop_keep(hq[e], (unsigned)tl.row(r), drop_t) ? inv_keep : 0.f;
                                ds[e] = p[e] * (dp[e] * km - d4[e]);
                                p[e] *= km;
                            } else {
                                ds[e] = p[e] * (dp[e] - d4[e]);                // `scale` is applied once, to dK
                            }
                        }
                        if (row0 + 16 > rows) {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (row0 + 4 * fg + e >= rows) { p[e] = 0.f; ds[e] = 0.f; }
                        }
                        ph[r][hh] = pack4(p); dsh[r][hh] = pack4(ds);
                    }
                }
#pragma unroll
                for (int r = 0; r < R; ++r) { pb[r] = cat8(ph[r][0], ph[r][1]); dsb[r] = cat8(dsh[r][0], dsh[r][1]); }
            }
            s16x4 qlo[NFD], qhi[NFD];
            {
                s16x4 lo[NFD], hi[NFD];
                tr_read_all<NFD, LD>(lo, hi, lds_addr_of(Ds) + troff + s * (32 * LD));
                tr_wait<NFD>(lo, hi);
                // the Q^T fragments of the dK product: in flight under the dV MFMAs
                if constexpr (VL_EARLY_TR) tr_read_all<NFD, LD>(qlo, qhi, lds_addr_of(Qs) + troff + s * (32 * LD));
#pragma unroll
                for (int fd = 0; fd < NFD; ++fd) {
                    const bf16x8 dt = tr_join(lo[fd], hi[fd]);
#pragma unroll
                    for (int r = 0; r < R; ++r) accV[r][fd] = MFMA(dt, pb[r], accV[r][fd]);       // dV^T[d][key]
                }
            }
            {
                s16x4 (&lo)[NFD] = qlo; s16x4 (&hi)[NFD] = qhi;
                if constexpr (!VL_EARLY_TR) tr_read_all<NFD, LD>(lo, hi, lds_addr_of(Qs) + troff + s * (32 * LD));
                tr_wait<NFD>(lo, hi);
#pragma unroll
                for (int fd = 0; fd < NFD; ++fd) {
                    const bf16x8 qt = tr_join(lo[fd], hi[fd]);
#pragma unroll
                    for (int r = 0; r < R; ++r) accK[r][fd] = MFMA(qt, dsb[r], accK[r][fd]);      // dK^T[d][key]
                }
            }
        }
    }
    if (!wave_active) return;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const bool ok = tl.row(r) < nk;
        store_rows16<NFD>(dk.p + (long long)tl.grow(r) * dk.s_n + h * dk.s_h, accK[r], scale, fg, ok);
        store_rows16<NFD>(dv.p + (long long)tl.grow(r) * dv.s_n + h * dv.s_h, accV[r], 1.f, fg, ok);
    }
}

HND to_hnd(vitk_hnd t) { return HND{(__bf16*)t.p, (long long)t.s_h, (long long)t.s_n}; }
bool hnd_ok(vitk_hnd t) { return t.p && aligned16(t.p) && (t.s_h % 8 == 0) && (t.s_n % 8 == 0); }

// geometry of the launches: 2 = (R, NW) = (2, 4), the default; 1 = (1, 8).  VITK_ATTN_VL overrides (A/B runs).
// experiments (VITK_VL_DBG): bit 0 = no LDS-DMA after the first chunk, bit 1 = no arithmetic, bit 2 = no barrier after the first (results are wrong)
// bits 8..: order of the workgroups (vl_block_of), VITK_VL_ORDER
int vl_dbg() {
    static const int g = [] { const char* e = vitk_exp("VITK_VL_DBG"); const char* o = vitk_exp("VITK_VL_ORDER");
                              return (e ? atoi(e) & 255 : 0) | ((o ? atoi(o) : 2) << 8); }();
    return g;
}
// one workgroup per (block, head), rounded up to a multiple of the 8 XCDs (vl_block_of retires the surplus)
unsigned vl_grid(int64_t nblk, int64_t H) { return (unsigned)((nblk * H + 7) / 8 * 8); }
// geometry of the launches: 2 = (R, NW) = (2, 4), 1 = (1, 8).  [measured, profiles/r04f_vl_orders.log] d = 80, N = 577: (2, 4) is 6 % ahead
// in the backward (2,339 vs 2,482 us at batch 256) and level in the forward; d = 64 (NaViT mix): (1, 8) is 6 % ahead in the forward and
// level in the backward.  VITK_ATTN_VL = 1 / 2 overrides (A/B runs).
int vl_geometry(int64_t d) {
    static const int g = [] { const char* e = vitk_exp("VITK_ATTN_VL"); return e ? atoi(e) : 0; }();
    return g == 1 || g == 2 ? g : (d > 64 ? 2 : 1);
}

}  // namespace

extern "C" int vitk_attn_varlen_fwd_bf16(vitk_hnd q, vitk_hnd k, vitk_hnd v, vitk_hnd o, float* lse, const int32_t* cu_q,
                                         const int32_t* cu_k, const int32_t* blk_seg, const int32_t* blk_r0, int64_t nblk,
                                         int64_t tq_total, int64_t H, int64_t d, float scale, void* stream) {
    return vitk_attn_varlen_fwd_bf16_drop(q, k, v, o, lse, cu_q, cu_k, blk_seg, blk_r0, nblk, tq_total, H, d, scale, 0.f, 0u, stream);
}

extern "C" int vitk_attn_varlen_fwd_bf16_drop(vitk_hnd q, vitk_hnd k, vitk_hnd v, vitk_hnd o, float* lse, const int32_t* cu_q,
                                              const int32_t* cu_k, const int32_t* blk_seg, const int32_t* blk_r0, int64_t nblk,
                                              int64_t tq_total, int64_t H, int64_t d, float scale, float drop_p, uint32_t drop_seed,
                                              void* stream) {
    if (!(drop_p >= 0.f && drop_p < 1.f)) VITK_FAIL(VITK_E_ARG, "attn_varlen_fwd_bf16: dropout p must be in [0, 1) (got %g)", (double)drop_p);
    if (H * tq_total > 0xffffffffLL) VITK_FAIL(VITK_E_SHAPE, "attn_varlen_fwd_bf16: H * tokens exceeds the 32-bit dropout row index");
    if (d != 32 && d != 48 && d != 64 && d != 80 && d != 96) VITK_FAIL(VITK_E_SHAPE, "attn_varlen_fwd_bf16: needs dim_head 32, 48, 64, 80 or 96 (got %lld)", (long long)d);
    if (!(scale > 0.f)) VITK_FAIL(VITK_E_ARG, "attn_varlen_fwd_bf16: scale must be positive (got %g)", (double)scale);
    if (!hnd_ok(q) || !hnd_ok(k) || !hnd_ok(v) || !hnd_ok(o) || !lse || !cu_q || !cu_k || !blk_seg || !blk_r0)
        VITK_FAIL(VITK_E_ALIGN, "attn_varlen_fwd_bf16: tensors must be non-null, 16-byte aligned with strides %% 8 == 0");
    if (nblk <= 0 || H <= 0 || H > 65535 || nblk * H > 0x7ffffff0LL) VITK_FAIL(VITK_E_SHAPE, "attn_varlen_fwd_bf16: empty problem or more than 2^31 workgroups");
#define VL_FWD_G(DHV, R_, NW_, DR_) hipLaunchKernelGGL((attn_varlen_fwd_kernel<DHV, R_, NW_, DR_>), dim3(vl_grid(nblk, H)), dim3(64 * NW_), 0, \
                       (hipStream_t)stream, to_hnd(q), to_hnd(k), to_hnd(v), to_hnd(o), lse, cu_q, cu_k, blk_seg, blk_r0, (int)tq_total, scale * LOG2E, \
                       drop_thresh(drop_p), drop_seed, 1.0f / (1.0f - drop_p), (int)nblk, (int)H, vl_dbg())
#define VL_FWD(DHV) do { const int g = vl_geometry(d); const bool dr = drop_thresh(drop_p) != 0u; \
    if (g == 2) { if (dr) VL_FWD_G(DHV, 2, 4, true); else VL_FWD_G(DHV, 2, 4, false); } else { if (dr) VL_FWD_G(DHV, 1, 8, true); else VL_FWD_G(DHV, 1, 8, false); } } while (0)
    switch ((int)d) {
        case 32: VL_FWD(32); break;
        case 48: VL_FWD(48); break;
        case 64: VL_FWD(64); break;
        case 80: VL_FWD(80); break;
        default: VL_FWD(96); break;
    }
#undef VL_FWD
#undef VL_FWD_G
    VITK_CHECK_LAUNCH("attn_varlen_fwd_bf16");
    return 0;
}

extern "C" int vitk_attn_varlen_bwd_bf16(vitk_hnd q, vitk_hnd k, vitk_hnd v, vitk_hnd o, vitk_hnd dout, const float* lse,
                                         float* delta, vitk_hnd dq, vitk_hnd dk, vitk_hnd dv, const int32_t* cu_q,
                                         const int32_t* cu_k, const int32_t* qblk_seg, const int32_t* qblk_r0, int64_t nqblk,
                                         const int32_t* kblk_seg, const int32_t* kblk_r0, int64_t nkblk, int64_t tq_total,
                                         int64_t H, int64_t d, float scale, void* stream) {
    return vitk_attn_varlen_bwd_bf16_drop(q, k, v, o, dout, lse, delta, dq, dk, dv, cu_q, cu_k, qblk_seg, qblk_r0, nqblk, kblk_seg, kblk_r0,
                                          nkblk, tq_total, H, d, scale, 0.f, 0u, stream);
}

extern "C" int vitk_attn_varlen_bwd_bf16_drop(vitk_hnd q, vitk_hnd k, vitk_hnd v, vitk_hnd o, vitk_hnd dout, const float* lse,
                                              float* delta, vitk_hnd dq, vitk_hnd dk, vitk_hnd dv, const int32_t* cu_q,
                                              const int32_t* cu_k, const int32_t* qblk_seg, const int32_t* qblk_r0, int64_t nqblk,
                                              const int32_t* kblk_seg, const int32_t* kblk_r0, int64_t nkblk, int64_t tq_total,
                                              int64_t H, int64_t d, float scale, float drop_p, uint32_t drop_seed, void* stream) {
    if (!(drop_p >= 0.f && drop_p < 1.f)) VITK_FAIL(VITK_E_ARG, "attn_varlen_bwd_bf16: dropout p must be in [0, 1) (got %g)", (double)drop_p);
    if (H * tq_total > 0xffffffffLL) VITK_FAIL(VITK_E_SHAPE, "attn_varlen_bwd_bf16: H * tokens exceeds the 32-bit dropout row index");
    const unsigned drop_t = drop_thresh(drop_p);
    const float inv_keep = 1.0f / (1.0f - drop_p);
    if (d != 32 && d != 48 && d != 64 && d != 80 && d != 96) VITK_FAIL(VITK_E_SHAPE, "attn_varlen_bwd_bf16: needs dim_head 32, 48, 64, 80 or 96 (got %lld)", (long long)d);
    if (!(scale > 0.f)) VITK_FAIL(VITK_E_ARG, "attn_varlen_bwd_bf16: scale must be positive (got %g)", (double)scale);
    if (!hnd_ok(q) || !hnd_ok(k) || !hnd_ok(v) || !hnd_ok(o) || !hnd_ok(dout) || !hnd_ok(dq) || !hnd_ok(dk) || !hnd_ok(dv) || !lse ||
        !delta || !cu_q || !cu_k || !qblk_seg || !qblk_r0 || !kblk_seg || !kblk_r0)
        VITK_FAIL(VITK_E_ALIGN, "attn_varlen_bwd_bf16: tensors must be non-null, 16-byte aligned with strides %% 8 == 0");
    if (nqblk <= 0 || nkblk <= 0 || H <= 0 || H > 65535 || nqblk * H > 0x7ffffff0LL || nkblk * H > 0x7ffffff0LL)
        VITK_FAIL(VITK_E_SHAPE, "attn_varlen_bwd_bf16: empty problem or more than 2^31 workgroups");
    hipStream_t st = (hipStream_t)stream;
#define VL_BWD_G(DHV, R_, NW_, DR_) do { \
    hipLaunchKernelGGL((attn_varlen_bwd_dq_kernel<DHV, R_, NW_, DR_>), dim3(vl_grid(nqblk, H)), dim3(64 * NW_), 0, st, to_hnd(q), to_hnd(k), \
                       to_hnd(v), to_hnd(o), to_hnd(dout), lse, delta, to_hnd(dq), cu_q, cu_k, qblk_seg, qblk_r0, (int)tq_total, scale, drop_t, drop_seed, inv_keep, (int)nqblk, (int)H, vl_dbg()); \
    hipLaunchKernelGGL((attn_varlen_bwd_dkv_kernel<DHV, R_, NW_, DR_>), dim3(vl_grid(nkblk, H)), dim3(64 * NW_), 0, st, to_hnd(q), to_hnd(k), \
                       to_hnd(v), to_hnd(dout), lse, delta, to_hnd(dk), to_hnd(dv), cu_q, cu_k, kblk_seg, kblk_r0, (int)tq_total, scale, drop_t, drop_seed, inv_keep, (int)nkblk, (int)H, vl_dbg()); } while (0)
#define VL_BWD(DHV) do { const int g = vl_geometry(d); const bool dr = drop_thresh(drop_p) != 0u; \
    if (g == 2) { if (dr) VL_BWD_G(DHV, 2, 4, true); else VL_BWD_G(DHV, 2, 4, false); } else { if (dr) VL_BWD_G(DHV, 1, 8, true); else VL_BWD_G(DHV, 1, 8, false); } } while (0)
    switch ((int)d) {
        case 32: VL_BWD(32); break;
        case 48: VL_BWD(48); break;
        case 64: VL_BWD(64); break;
        case 80: VL_BWD(80); break;
        default: VL_BWD(96); break;
    }
#undef VL_BWD
#undef VL_BWD_G
    VITK_CHECK_LAUNCH("attn_varlen_bwd_dkv");
    return 0;
}
