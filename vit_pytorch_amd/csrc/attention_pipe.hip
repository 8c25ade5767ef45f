// attention_pipe.hip -- persistent, LDS-DMA-pipelined scaled-dot-product attention forward / backward for gfx950 (dim_head 64,
// 32 < N <= 224 tokens: every ViT at 224^2 with patch 16 -- N = 197 / 196).
//
// Replaces vit.py:55-63 (simple_vit.py:54-61) and its autograd, like attention.hip, with the same fragment algebra (S^T = K Q^T,
// P feeding P.V straight from the accumulators, lazy reference maximum, MFMA row sums; backward as a query-tile-outer dQ kernel
// and a key-tile-outer dK/dV kernel that recompute P from the saved log-sum-exp).  What is different is HOW THE OPERANDS ARRIVE:
//
//   attention.hip     one workgroup per (batch, head): global -> VGPR -> ds_write of the whole head, barrier, compute.  Measured
//                     (DESIGN section 4): staging alone 34 us, compute alone 54 us, together 83 us -- the two never overlap, because
//                     every workgroup of a CU is in the same phase and nothing is in flight while a head is being multiplied.
//   here              RESIDENT workgroups walk a list of (batch, head) items.  One PRODUCER wave per workgroup copies the NEXT
//                     operands HBM -> LDS with global_load_lds (no VGPR round trip, no ds_write issue, nothing the compute waves
//                     wait for) while the other waves multiply the CURRENT ones:
//                       forward   2 workgroups / CU x 8 waves, ONE K|V buffer per workgroup split into two halves of the key
//                                 range; half h + 1 is in flight while half h is multiplied (the online softmax walks the keys
//                                 once, so a half is dead as soon as it has been read) -- 57 KB per workgroup;
//                       backward  1 workgroup / CU x 16 waves, TWO whole-head buffers (K|V for dQ, Q|dO|lse|delta for dK/dV):
//                                 item i + 1 loads while item i is multiplied -- 115-119 KB.
//                     The rows each wave owns (its query / key tile) are fetched right after the wave's last step, before its
//                     stores, so they are in flight across the epilogue and the item barrier.
//
// LDS image: 128-byte rows (64 x 16 bit), no padding -- an LDS-DMA instruction writes lane-linearly (8 rows x 128 B per wave
// instruction), so the bank swizzle sits in the SOURCE address: position p of row r holds the row's 16-byte chunk p ^ (r & 7).
// Both fragment shapes are conflict-free on it: ds_read_b128 row fragments (16 rows x one chunk: the 16 lanes of a group land on
// 16 distinct 16-byte bank slots) and ds_read_b64_tr_b16 transpose fragments (8 rows x two chunks: 64 distinct banks).
//
// NS = 2 is the f32-ACCURATE flavour used by the f32 validation mode: every 16-bit operand arrives as hi + lo
// (vitk_split2), every product keeps hi.hi + hi.lo + lo.hi, the probabilities / dS are split the same way, outputs are f32:
// ~2^-16 per product instead of 2^-8, on the SAME staging, pipelining, masking and softmax code as the 16-bit kernels.
#include "common.h"
#include "attention_frag.h"
#include "attention_pipe.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

// The experiment bits of the kernels' argument structs (VITK_ATTN_DBG: ablations, cycle stamps) exist in the experiments flavour of the library
// only (round 6): in the product build they are the constant 0, so that the stamp code, its registers and its per-step branches -- basic-block
// boundaries in the middle of every step -- are not compiled into the kernels that run.
#ifdef VITK_EXPERIMENTS
#define AP_DBG(a) ((a).dbg)
#else
#define AP_DBG(a) 0
#endif

namespace {


struct TND { const __bf16* p; long long s_b, s_h, s_n; };      // 16-bit operand, element strides
struct OND { void* p; long long s_b, s_h, s_n; };              // output (16-bit, or f32 when NS == 2)

// workgroup barrier that also tells the COMPILER that LDS changed: the compute waves never write LDS themselves (the producer
// wave's DMA does), so without the clobber hipcc may keep or hoist fragment reads across items
#define AP_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); \
    __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); } while (0)
#define AP_WAIT_DMA() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

template <int NS> struct Fr { bf16x8 t[NS]; };
// a . b with the terms hi.hi (+ hi.lo + lo.hi): small terms first
template <int NS>
__device__ __forceinline__ f32x4 mm(const Fr<NS>& a, const Fr<NS>& b, f32x4 c) {
    if constexpr (NS == 2) { c = MFMA(a.t[1], b.t[0], c); c = MFMA(a.t[0], b.t[1], c); }
    return MFMA(a.t[0], b.t[0], c);
}
// two f32 fragments -> NS 16-bit terms
template <int NS>
__device__ __forceinline__ Fr<NS> split_pack(f32x4 a, f32x4 b) {
    Fr<NS> r;
    r.t[0] = pack8(a, b);
    if constexpr (NS == 2) {
        f32x4 ra, rb;
#pragma unroll
        for (int e = 0; e < 4; ++e) { ra[e] = a[e] - (float)r.t[0][e]; rb[e] = b[e] - (float)r.t[0][4 + e]; }
        r.t[1] = pack8(ra, rb);
    }
    return r;
}
// ---- swizzled LDS image -------------------------------------------------------------------------------------------------
// 16 rows x (32 of the 64 columns) as an MFMA A/B operand: lane (i = lane & 15, g = lane >> 4) holds tile[row0 + i][ks*32 + 8g .. +7]
__device__ __forceinline__ bf16x8 sw_row(const char* tile, int row0, int ks, int fi, int fg) {
    const int row = row0 + fi;
    return *reinterpret_cast<const bf16x8*>(tile + row * 128 + (((ks * 4 + fg) ^ (row & 7)) << 4));
}
// transposed operand: lane (i, g) holds tile[row0 + {4g..4g+3, 16+4g..16+4g+3}][col0 + i]   (col0 a multiple of 16)
__device__ __forceinline__ bf16x8 sw_tr(const char* tile, int row0, int col0, int fi, int fg) {
    const int row = row0 + 4 * fg + (fi >> 2);
    const char* p = tile + row * 128 + ((((col0 >> 3) + ((fi & 3) >> 1)) ^ (row & 7)) << 4) + ((fi & 1) << 3);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + 16 * 128));   // row + 16: same swizzle
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
template <int NS>
__device__ __forceinline__ Fr<NS> sw_row_n(const char* tile0, int term_stride, int row0, int ks, int fi, int fg) {
    Fr<NS> r;
#pragma unroll
    for (int t = 0; t < NS; ++t) r.t[t] = sw_row(tile0 + t * term_stride, row0, ks, fi, fg);
    return r;
}
template <int NS>
__device__ __forceinline__ Fr<NS> sw_tr_n(const char* tile0, int term_stride, int row0, int col0, int fi, int fg) {
    Fr<NS> r;
#pragma unroll
    for (int t = 0; t < NS; ++t) r.t[t] = sw_tr(tile0 + t * term_stride, row0, col0, fi, fg);
    return r;
}
// producer: row groups [g0, g1) (8 rows each) of one (n, 64) head slice -> LDS tile; rows past N re-read row N - 1, so every LDS
// row always holds finite data (padding keys are masked by value, padding queries by index)
__device__ __forceinline__ void dma_rows(const __bf16* base, long long s_n, int N, char* tile, int g0, int g1, int lane) {
    const int lrow = lane >> 3, lchunk = (lane & 7) ^ (lane >> 3);       // LDS row 8g + lrow, position lane & 7 <- chunk lchunk
    for (int g = g0; g < g1; ++g) {
        int row = 8 * g + lrow; row = row < N ? row : N - 1;
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(base + (long long)row * s_n + lchunk * 8),
                                         (void __attribute__((address_space(3)))*)(tile + g * 1024), 16, 0, 0);
    }
}
// producer: n floats (n <= 256) -> LDS, indices past N re-read element N - 1
__device__ __forceinline__ void dma_f32(const float* base, int N, char* dst, int n, int lane) {
    for (int g = 0; g * 64 < n; ++g) {
        int i = g * 64 + lane; i = i < N ? i : N - 1;
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(base + i),
                                         (void __attribute__((address_space(3)))*)(dst + g * 256), 4, 0, 0);
    }
}
template <int NS>
__device__ __forceinline__ void load_row_frags(Fr<NS> (&f)[2], const TND (&t)[NS], long long off, int fg) {
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const __bf16* p = t[i].p + off;
        f[0].t[i] = *reinterpret_cast<const bf16x8*>(p + 8 * fg);
        f[1].t[i] = *reinterpret_cast<const bf16x8*>(p + 32 + 8 * fg);
    }
}
template <int NS> using out_t = std::conditional_t<NS == 1, __bf16, float>;

// ==========================================================================================================================
// forward
// ==========================================================================================================================
template <int NS> struct FwdArgs {
    TND q[NS], k[NS], v[NS];
    OND o;
    float* lse;
    int H, N, nitems;
    float c;                 // scale * log2(e)
    unsigned drop_t, drop_seed;
    float inv_keep;
    int dbg;                 // experiments (VITK_ATTN_DBG): bit 0 = producer issues no DMA, bit 1 = compute waves skip the steps
};

// 8 waves: 0-6 carry two 16-row query tiles each (N <= 224), wave 7 is the producer.  Two workgroups per CU (NS = 1).
template <int NS, bool DROP>
__global__ __launch_bounds__(512, NS == 1 ? 4 : 2) void attn_fwd_pipe_kernel(const FwdArgs<NS> a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = a.N, H = a.H;
    const int nks = (N + 31) >> 5, nA = (nks + 1) >> 1;
    const int tile_b = nks * 32 * 128;              // one tensor, one term
    const int term_b = 2 * tile_b;                  // K | V of one term
    char* const Ks = smem;
    char* const Vs = smem + tile_b;
    const int first = blockIdx.x, stride = gridDim.x;

    if (wave == 7) {
        // ---- producer: half h + 1 of the key range is in flight while the compute waves multiply half h ----
        for (int item = first; item < a.nitems; item += stride) {
            const int b = item / H, h = item - b * H;
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                const int g0 = part ? 4 * nA : 0, g1 = part ? 4 * nks : 4 * nA;
                if (!(a.dbg & 1)) {
#pragma unroll
                    for (int t = 0; t < NS; ++t) {
                        dma_rows(a.k[t].p + b * a.k[t].s_b + h * a.k[t].s_h, a.k[t].s_n, N, Ks + t * term_b, g0, g1, lane);
                        dma_rows(a.v[t].p + b * a.v[t].s_b + h * a.v[t].s_h, a.v[t].s_n, N, Vs + t * term_b, g0, g1, lane);
                    }
                }
                AP_WAIT_DMA();
                AP_BARRIER();       // publishes this half; everyone has finished the half this buffer held before the PREVIOUS barrier
            }
        }
        return;
    }

    // ---- compute waves ----
    using TO = out_t<NS>;
    const int fi = lane & 15, fg = lane >> 4;
    const int nqt = (N + 15) >> 4;
    const int t0 = wave * 2;
    const bool active = t0 < nqt;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const bf16x8 ones = {(__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f};
    const float c = a.c;
    Fr<NS> qf[2][2];
    auto load_q = [&](int item) {
        const int b = item / H, h = item - b * H;
        int lane_o = lane;                             // opaque copy: the lane offsets are re-derived per item -- hoisted to kernel entry they
        asm volatile("" : "+v"(lane_o));               // are spilled around the item loop (and a reload draws a vmcnt(0) between the loads)
        const int fi_ = lane_o & 15, fg_ = lane_o >> 4;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int qi = (t0 + r) * 16 + fi_;
            load_row_frags<NS>(qf[r], a.q, b * a.q[0].s_b + h * a.q[0].s_h + (long long)(qi < N ? qi : N - 1) * a.q[0].s_n, fg_);
        }
    };
    if (active && first < a.nitems) load_q(first);

    for (int item = first; item < a.nitems; item += stride) {
        const int b = item / H, h = item - b * H;
        float mref[2];
        f32x4 acc[2][4], accl[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) { mref[r] = -INFINITY; accl[r] = z4; acc[r][0] = acc[r][1] = acc[r][2] = acc[r][3] = z4; }

        auto step = [&](int s) {
            Fr<NS> kf[2][2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                kf[hh][0] = sw_row_n<NS>(Ks, term_b, s * 32 + hh * 16, 0, fi, fg);
                kf[hh][1] = sw_row_n<NS>(Ks, term_b, s * 32 + hh * 16, 1, fi, fg);
            }
            Fr<NS> pb[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                f32x4 st[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    st[hh] = mm<NS>(kf[hh][0], qf[r][0], z4);
                    st[hh] = mm<NS>(kf[hh][1], qf[r][1], st[hh]);
                }
                if (s == nks - 1) {            // only the last 32-key step holds padding keys
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (s * 32 + hh * 16 + 4 * fg + e >= N) st[hh][e] = -INFINITY;
                }
                // lazy reference maximum, MFMA row sums, scale and -mref folded into the exp2 argument: see attention.hip
                float mloc = fmaxf(fmaxf(st[0][0], st[0][1]), st[0][2]);
                mloc = fmaxf(fmaxf(mloc, st[0][3]), st[1][0]);
                mloc = fmaxf(fmaxf(mloc, st[1][1]), st[1][2]);
                mloc = fmaxf(mloc, st[1][3]);
                if (__builtin_amdgcn_ballot_w64(mloc * c > mref[r] + 8.0f) != 0) {
                    const float m_new = fmaxf(mref[r], groups_max(mloc) * c);
                    const float alpha = __builtin_amdgcn_exp2f(mref[r] - m_new);
#pragma unroll
                    for (int fd = 0; fd < 4; ++fd) acc[r][fd] *= alpha;
                    accl[r] *= alpha;
                    mref[r] = m_new;
                }
                const float nm = -mref[r];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                    for (int e = 0; e < 4; ++e) st[hh][e] = __builtin_amdgcn_exp2f(fmaf(st[hh][e], c, nm));
                pb[r] = split_pack<NS>(st[0], st[1]);
#pragma unroll
                for (int t = 0; t < NS; ++t) accl[r] = MFMA(ones, pb[r].t[t], accl[r]);      // softmax denominators: of the UNDROPPED probabilities
                if constexpr (DROP) {                          // nn.Dropout on the attention matrix (vit.py:60): zero P entries for P.V
                    const unsigned hrow = drop_row((unsigned)(item * N + (t0 + r) * 16 + fi), a.drop_seed);
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (!drop_keep(hrow, (unsigned)(s * 32 + hh * 16 + 4 * fg + e), a.drop_t)) st[hh][e] = 0.f;
                    pb[r] = split_pack<NS>(st[0], st[1]);
                }
            }
#pragma unroll
            for (int fd = 0; fd < 4; ++fd) {
                const Fr<NS> vf = sw_tr_n<NS>(Vs, term_b, s * 32, fd * 16, fi, fg);
#pragma unroll
                for (int r = 0; r < 2; ++r) acc[r][fd] = mm<NS>(vf, pb[r], acc[r][fd]);
            }
        };

        AP_BARRIER();                                  // first half of the keys has landed
        const bool work = active && !(a.dbg & 2);
        if (work) for (int s = 0; s < nA; ++s) step(s);
        AP_BARRIER();                                  // second half has landed (and the first is free for the next item)
        if (active) {
            if (work) for (int s = nA; s < nks; ++s) step(s);
            const int nxt = item + stride;
            if (nxt < a.nitems) load_q(nxt);           // in flight across the stores below and the next item's barrier
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int qi = (t0 + r) * 16 + fi;
                const float ls = accl[r][0];           // every row of 1^T P^T is the same sum
                const float inv = a.inv_keep / ls;     // kept entries are scaled by 1 / (1 - p)
                if (qi < N) {
                    TO* op = reinterpret_cast<TO*>(a.o.p) + (b * a.o.s_b + h * a.o.s_h) + (qi * (int)a.o.s_n + 4 * fg);
#pragma unroll
                    for (int fd = 0; fd < 4; ++fd) store4<TO>(op + fd * 16, acc[r][fd] * inv);
                    if (fg == 0) a.lse[(long long)item * N + qi] = (mref[r] + log2f(ls)) * LN2;
                }
            }
        }
    }
}

// ==========================================================================================================================
// backward, dQ: query-tile outer; K | V of the whole head double-buffered in LDS (NS = 1)
// ==========================================================================================================================
template <int NS> struct DqArgs {
    TND q[NS], k[NS], v[NS], dout[NS];
    OND o;                   // the forward's output: 16-bit (NS = 1) or f32 (NS = 2)
    const float* lse;
    float* delta;
    OND dq;
    int H, N, nitems;
    float scale;
    unsigned drop_t, drop_seed;
    float inv_keep;
    int dbg;
};

// 16 waves: 0-14 carry one 16-row tile each (N <= 224: 14 tiles), wave 15 is the producer.  One workgroup per CU.
template <int NS, bool DROP>
__global__ __launch_bounds__(1024) void attn_dq_pipe_kernel(const DqArgs<NS> a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NBUF = NS == 1 ? 2 : 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = a.N, H = a.H;
    const int nks = (N + 31) >> 5;
    const int tile_b = nks * 32 * 128, term_b = 2 * tile_b, buf_b = NS * term_b;
    const int first = blockIdx.x, stride = gridDim.x;

    if (wave == 15) {
        auto issue = [&](int item, int buf) {
            if (a.dbg & 1) return;
            const int b = item / H, h = item - b * H;
            char* base = smem + buf * buf_b;
#pragma unroll
            for (int t = 0; t < NS; ++t) {
                dma_rows(a.k[t].p + b * a.k[t].s_b + h * a.k[t].s_h, a.k[t].s_n, N, base + t * term_b, 0, 4 * nks, lane);
                dma_rows(a.v[t].p + b * a.v[t].s_b + h * a.v[t].s_h, a.v[t].s_n, N, base + t * term_b + tile_b, 0, 4 * nks, lane);
            }
        };
        if constexpr (NBUF == 2) {
            int j = 0;
            if (first < a.nitems) issue(first, 0);
            for (int item = first; item < a.nitems; item += stride, ++j) {
                AP_WAIT_DMA();
                AP_BARRIER();                          // item's buffer is published; everyone has left the other buffer
                if (item + stride < a.nitems) issue(item + stride, (j + 1) & 1);
            }
        } else {
            for (int item = first; item < a.nitems; item += stride) {
                issue(item, 0);
                AP_WAIT_DMA();
                AP_BARRIER();
                AP_BARRIER();                          // everyone has finished reading
            }
        }
        return;
    }

    using TO = out_t<NS>;
    const int fi = lane & 15, fg = lane >> 4;
    const int nqt = (N + 15) >> 4;
    const bool active = wave < nqt;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const float c = a.scale * LOG2E;
    const int qi = wave * 16 + fi;
    const int qrow = qi < N ? qi : N - 1;

    Fr<NS> qf[2], df[2];
    float l2 = 0.f, dl = 0.f;
    // the rows this wave owns: q, dO, the forward's output (for delta = rowsum(dO * O)), lse
    struct Rows { Fr<NS> q[2], d[2]; std::conditional_t<NS == 1, bf16x8, f32x4> o[NS == 1 ? 2 : 4]; float lse; };
    auto fetch_rows = [&](int item) {
        Rows r;
        const int b = item / H, h = item - b * H;
        load_row_frags<NS>(r.q, a.q, b * a.q[0].s_b + h * a.q[0].s_h + (long long)qrow * a.q[0].s_n, fg);
        load_row_frags<NS>(r.d, a.dout, b * a.dout[0].s_b + h * a.dout[0].s_h + (long long)qrow * a.dout[0].s_n, fg);
        r.lse = a.lse[(long long)item * N + qrow];
        const TO* op = reinterpret_cast<const TO*>(a.o.p) + b * a.o.s_b + h * a.o.s_h + (long long)qrow * a.o.s_n;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if constexpr (NS == 1) r.o[ks] = *reinterpret_cast<const bf16x8*>(op + ks * 32 + 8 * fg);
            else { r.o[2 * ks] = *reinterpret_cast<const f32x4*>(op + ks * 32 + 8 * fg); r.o[2 * ks + 1] = *reinterpret_cast<const f32x4*>(op + ks * 32 + 8 * fg + 4); }
        }
        return r;
    };
    // first use of the rows (this is where the loads are waited for): delta = rowsum(dO * O), this lane's 16 columns
    auto adopt = [&](const Rows& r) {
        qf[0] = r.q[0]; qf[1] = r.q[1]; df[0] = r.d[0]; df[1] = r.d[1];
        l2 = -r.lse * LOG2E;                                 // negated: the exp2 argument is one FMA
        float s = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float d = (float)r.d[ks].t[0][e];
                if constexpr (NS == 2) d += (float)r.d[ks].t[1][e];
                float o;
                if constexpr (NS == 1) o = (float)r.o[ks][e]; else o = r.o[2 * ks + (e >> 2)][e & 3];
                s = fmaf(d, o, s);
            }
        dl = s;                                              // summed over the 4 lane groups at use
    };
    if (active && first < a.nitems) {
        adopt(fetch_rows(first));
        // consume every row register HERE: left outstanding, the first item's loads make hipcc's merged counter state at the loop
        // header "loads in flight" and it then waits vmcnt(0) at every item start -- i.e. for the previous item's stores
        if constexpr (NS == 1) asm volatile("" :: "v"(qf[0].t[0]), "v"(qf[1].t[0]), "v"(df[0].t[0]), "v"(df[1].t[0]), "v"(l2), "v"(dl));
    }

    int j = 0;
    for (int item = first; item < a.nitems; item += stride, ++j) {
        const int b = item / H, h = item - b * H;
        const char* Ks = smem + (NBUF == 2 ? (j & 1) * buf_b : 0);
        const char* Vs = Ks + tile_b;
        AP_BARRIER();
        if (active) {
            const float dsum = groups_sum(dl);
            if (qi < N && fg == 0) a.delta[(long long)item * N + qi] = dsum;
            // NS = 1: the next item's rows are requested NOW, into their own registers, and adopted after the last step: the
            // loads have the whole item to land (fetched after the last step they cost a memory round trip per item -- and,
            // vmcnt retiring in order, the completion of this item's stores on top of it)
            // (UNCONDITIONAL, the last item re-fetches its own rows: under `if (more)` hipcc cannot pair the fetch with the adoption
            //  below, assumes loads may be pending at the loop header and waits for this item's STORES before it re-uses a register)
            const int nitem = item + stride < a.nitems ? item + stride : item;
            Rows nxt;
            if constexpr (NS == 1) nxt = fetch_rows(nitem);
            f32x4 acc[4];
            acc[0] = acc[1] = acc[2] = acc[3] = z4;
            for (int s = 0; s < ((a.dbg & 2) ? 0 : nks); ++s) {
                f32x4 ds[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int row0 = s * 32 + hh * 16;
                    const Fr<NS> k0 = sw_row_n<NS>(Ks, term_b, row0, 0, fi, fg), k1 = sw_row_n<NS>(Ks, term_b, row0, 1, fi, fg);
                    const Fr<NS> v0 = sw_row_n<NS>(Vs, term_b, row0, 0, fi, fg), v1 = sw_row_n<NS>(Vs, term_b, row0, 1, fi, fg);
                    f32x4 st = mm<NS>(k0, qf[0], z4);
                    st = mm<NS>(k1, qf[1], st);
                    f32x4 dp = mm<NS>(v0, df[0], z4);
                    dp = mm<NS>(v1, df[1], dp);
                    if constexpr (DROP) {                   // dP = dP_dropped * keep / (1 - p)
                        const unsigned hrow = drop_row((unsigned)(item * N + wave * 16 + fi), a.drop_seed);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            dp[e] = drop_keep(hrow, (unsigned)(row0 + 4 * fg + e), a.drop_t) ? dp[e] * a.inv_keep : 0.f;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e)           // the common factor `scale` of dS is applied once, to dQ
                        ds[hh][e] = __builtin_amdgcn_exp2f(fmaf(st[e], c, l2)) * (dp[e] - dsum);
                    if (s == nks - 1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (row0 + 4 * fg + e >= N) ds[hh][e] = 0.f;
                    }
                }
                const Fr<NS> dsb = split_pack<NS>(ds[0], ds[1]);
#pragma unroll
                for (int fd = 0; fd < 4; ++fd) acc[fd] = mm<NS>(sw_tr_n<NS>(Ks, term_b, s * 32, fd * 16, fi, fg), dsb, acc[fd]);
            }
            if constexpr (NS != 1) nxt = fetch_rows(nitem);
            adopt(nxt);
            // pin the adoption (and with it the wait for the row loads) IN FRONT of this item's stores: hipcc otherwise sinks it
            // below them and its vmcnt(0) then waits for the stores' write round trip
            if constexpr (NS == 1) asm volatile("" :: "v"(qf[0].t[0]), "v"(qf[1].t[0]), "v"(df[0].t[0]), "v"(df[1].t[0]), "v"(l2), "v"(dl) : "memory");
            if constexpr (NS == 1) {
                store_rows16<4>(reinterpret_cast<__bf16*>(a.dq.p) + (b * a.dq.s_b + h * a.dq.s_h) + (qi < N ? qi : 0) * (int)a.dq.s_n, acc, a.scale, fg, qi < N);
            } else if (qi < N) {
                TO* dqp = reinterpret_cast<TO*>(a.dq.p) + (b * a.dq.s_b + h * a.dq.s_h) + (qi * (int)a.dq.s_n + 4 * fg);
#pragma unroll
                for (int fd = 0; fd < 4; ++fd) store4<TO>(dqp + fd * 16, acc[fd] * a.scale);
            }
        }
        if constexpr (NBUF == 1) AP_BARRIER();
    }
}

// ==========================================================================================================================
// backward, dK / dV: key-tile outer; Q | dO (+ lse, delta) of the whole head double-buffered in LDS (NS = 1)
// ==========================================================================================================================
template <int NS> struct DkvArgs {
    TND q[NS], k[NS], v[NS], dout[NS];
    const float* lse;
    const float* delta;
    OND dk, dv;
    int H, N, nitems;
    float scale;
    unsigned drop_t, drop_seed;
    float inv_keep;
    int dbg;
};

template <int NS, bool DROP>
__global__ __launch_bounds__(1024) void attn_dkv_pipe_kernel(const DkvArgs<NS> a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NBUF = NS == 1 ? 2 : 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = a.N, H = a.H;
    const int nqs = (N + 31) >> 5;
    const int tile_b = nqs * 32 * 128, term_b = 2 * tile_b, buf_b = NS * term_b + 2048;     // + lse, delta: 256 floats each
    const int first = blockIdx.x, stride = gridDim.x;

    if (wave == 15) {
        auto issue = [&](int item, int buf) {
            if (a.dbg & 1) return;
            const int b = item / H, h = item - b * H;
            char* base = smem + buf * buf_b;
#pragma unroll
            for (int t = 0; t < NS; ++t) {
                dma_rows(a.q[t].p + b * a.q[t].s_b + h * a.q[t].s_h, a.q[t].s_n, N, base + t * term_b, 0, 4 * nqs, lane);
                dma_rows(a.dout[t].p + b * a.dout[t].s_b + h * a.dout[t].s_h, a.dout[t].s_n, N, base + t * term_b + tile_b, 0, 4 * nqs, lane);
            }
            dma_f32(a.lse + (long long)item * N, N, base + NS * term_b, nqs * 32, lane);
            dma_f32(a.delta + (long long)item * N, N, base + NS * term_b + 1024, nqs * 32, lane);
        };
        if constexpr (NBUF == 2) {
            int j = 0;
            if (first < a.nitems) issue(first, 0);
            for (int item = first; item < a.nitems; item += stride, ++j) {
                AP_WAIT_DMA();
                AP_BARRIER();
                if (item + stride < a.nitems) issue(item + stride, (j + 1) & 1);
            }
        } else {
            for (int item = first; item < a.nitems; item += stride) {
                issue(item, 0);
                AP_WAIT_DMA();
                AP_BARRIER();
                AP_BARRIER();
            }
        }
        return;
    }

    using TO = out_t<NS>;
    const int fi = lane & 15, fg = lane >> 4;
    const int nkt = (N + 15) >> 4;
    const bool active = wave < nkt;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const float c = a.scale * LOG2E;
    const int ki = wave * 16 + fi;
    const int krow = ki < N ? ki : N - 1;

    Fr<NS> kf[2], vf[2];
    auto fetch_rows = [&](int item) {
        const int b = item / H, h = item - b * H;
        load_row_frags<NS>(kf, a.k, b * a.k[0].s_b + h * a.k[0].s_h + (long long)krow * a.k[0].s_n, fg);
        load_row_frags<NS>(vf, a.v, b * a.v[0].s_b + h * a.v[0].s_h + (long long)krow * a.v[0].s_n, fg);
    };
    if (active && first < a.nitems) fetch_rows(first);

    int j = 0;
    for (int item = first; item < a.nitems; item += stride, ++j) {
        const int b = item / H, h = item - b * H;
        const char* Qs = smem + (NBUF == 2 ? (j & 1) * buf_b : 0);
        const char* Ds = Qs + tile_b;
        const float* lse_s = reinterpret_cast<const float*>(Qs + NS * term_b);
        const float* del_s = lse_s + 256;
        AP_BARRIER();
        if (active) {
            f32x4 accK[4], accV[4];
#pragma unroll
            for (int fd = 0; fd < 4; ++fd) { accK[fd] = z4; accV[fd] = z4; }
            for (int s = 0; s < ((a.dbg & 2) ? 0 : nqs); ++s) {
                f32x4 p[2], ds[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int row0 = s * 32 + hh * 16;
                    const Fr<NS> q0 = sw_row_n<NS>(Qs, term_b, row0, 0, fi, fg), q1 = sw_row_n<NS>(Qs, term_b, row0, 1, fi, fg);
                    const Fr<NS> d0 = sw_row_n<NS>(Ds, term_b, row0, 0, fi, fg), d1 = sw_row_n<NS>(Ds, term_b, row0, 1, fi, fg);
                    const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + row0 + 4 * fg);
                    const f32x4 d4 = *reinterpret_cast<const f32x4*>(del_s + row0 + 4 * fg);
                    f32x4 st = mm<NS>(q0, kf[0], z4);      // S[q = row0 + 4g + e][key]
                    st = mm<NS>(q1, kf[1], st);
                    f32x4 dp = mm<NS>(d0, vf[0], z4);      // dP[q][key]
                    dp = mm<NS>(d1, vf[1], dp);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        p[hh][e] = __builtin_amdgcn_exp2f(fmaf(st[e], c, -LOG2E * l4[e]));
                        if constexpr (DROP) {              // dV takes P * keep / (1 - p); dP = dP_dropped * keep / (1 - p)
                            const unsigned hq = drop_row((unsigned)(item * N + row0 + 4 * fg + e), a.drop_seed);
                            const float km = drop_keep(hq, (unsigned)(wave * 16 + fi), a.drop_t) ? a.inv_keep : 0.f;
                            ds[hh][e] = p[hh][e] * (dp[e] * km - d4[e]);
                            p[hh][e] *= km;
                        } else {
                            ds[hh][e] = p[hh][e] * (dp[e] - d4[e]);      // `scale` is applied once, to dK
                        }
                    }
                    if (s == nqs - 1) {                    // padding query rows only exist in the last step
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (row0 + 4 * fg + e >= N) { p[hh][e] = 0.f; ds[hh][e] = 0.f; }
                    }
                }
                const Fr<NS> pb = split_pack<NS>(p[0], p[1]), dsb = split_pack<NS>(ds[0], ds[1]);
#pragma unroll
                for (int fd = 0; fd < 4; ++fd) {
                    accV[fd] = mm<NS>(sw_tr_n<NS>(Ds, term_b, s * 32, fd * 16, fi, fg), pb, accV[fd]);    // dV^T[d][key]
                    accK[fd] = mm<NS>(sw_tr_n<NS>(Qs, term_b, s * 32, fd * 16, fi, fg), dsb, accK[fd]);   // dK^T[d][key]
                }
            }
            if (item + stride < a.nitems) fetch_rows(item + stride);     // (an earlier fetch into spare registers, as in the dQ kernel, does not fit in 128 VGPRs here)
            if constexpr (NS == 1) {
                store_rows16<4>(reinterpret_cast<__bf16*>(a.dk.p) + (b * a.dk.s_b + h * a.dk.s_h) + (ki < N ? ki : 0) * (int)a.dk.s_n, accK, a.scale, fg, ki < N);
                store_rows16<4>(reinterpret_cast<__bf16*>(a.dv.p) + (b * a.dv.s_b + h * a.dv.s_h) + (ki < N ? ki : 0) * (int)a.dv.s_n, accV, 1.f, fg, ki < N);
            } else if (ki < N) {
                TO* dkp = reinterpret_cast<TO*>(a.dk.p) + (b * a.dk.s_b + h * a.dk.s_h) + (ki * (int)a.dk.s_n + 4 * fg);
                TO* dvp = reinterpret_cast<TO*>(a.dv.p) + (b * a.dv.s_b + h * a.dv.s_h) + (ki * (int)a.dv.s_n + 4 * fg);
#pragma unroll
                for (int fd = 0; fd < 4; ++fd) { store4<TO>(dkp + fd * 16, accK[fd] * a.scale); store4<TO>(dvp + fd * 16, accV[fd]); }
            }
        }
        if constexpr (NBUF == 1) AP_BARRIER();
    }
}

// ==========================================================================================================================
// backward, FUSED: dQ, dK, dV of one (batch, head) item in ONE pass over the scores (16-bit operands, 192 < N <= 208: the ViT-B/L
// sequence lengths at 224^2 / patch 16).  Why: the two-kernel backward computes S and exp twice (7 matmul units instead of 5), reads
// q, k, v, dO twice (927 MB instead of 618 MB at ViT-B/16 batch 256), and -- measured, see DESIGN_HISTORY.md, round 3 -- spends half
// its time on the rows every wave fetches for itself at the start of an item.  Here NO compute wave loads from global memory
// (KT = 1 described, the flavour in use):
//
//   waves 0-12   "KV waves": wave w owns key tile w (16 keys): K / V fragments from LDS at the start of the item, then per 32-query
//                step  S = Q K^T, dP = dO V^T - delta (delta enters as the accumulator's start), P = exp2(S c - lse log2 e),
//                dS = P dP  ->  dV += P^T dO, dK += dS^T Q  (accumulated in registers over the 7 steps) and dS^T (16 bit) -> LDS;
//   waves 13-14  "dQ waves": one step behind, dQ[32 queries] = dS K over ALL keys from the LDS copy of dS^T and the resident K
//                (transpose reads, K^T of five key steps held in registers for the item; wave 13 the first 32 columns of d, wave 14
//                the rest), left in LDS for the producer to store; their only global accesses are the O rows for
//                delta = rowsum(dO * O) of the step AHEAD (dO from the ring), which hipcc counts exactly; -delta and -lse log2 e
//                go to LDS for the KV waves;
//   wave 15      producer: Q | dO | lse of every 32-query step through a 5-stage LDS ring (global_load_lds; continuous across
//                items, four steps ahead), K of the NEXT item into the second K buffer and V of the next item into the single V
//                buffer (its fragments are read once, at the start of an item) during steps 1-4 of the current one; and the dQ
//                rows of step g - 2 from LDS to global memory as whole 128-byte rows, exactly four store instructions a step, so
//                that its vmcnt waits stay exact counts (9 ring pieces + 4 stores per step, + 13 K / V pieces in steps 1-4: the
//                operations of the last two steps may fly).
//   One workgroup barrier per step.  LDS: K 2 x 26 KB, V 26 KB, dS^T 2 x 14 KB, ring 5 x 8.25 KB, dQ 2 x 4 KB, delta + lse 512 B = 155.8 KB.
//
//   [measured, ViT-B/16 batch 256, one box; VITK_ATTN_DBG = 8 / 16 cycle stamps, tools/attn_ab.py]  A step takes ~4,100 cycles; every
//   wave waits >= 600 of them at the barrier and the waves with the highest ids on each SIMD arrive last: the four SIMDs are about
//   equally loaded and the step is bound by what a SIMD can issue, not by one role.  What moved it (216 -> 203 us): the producer's
//   addressing as uniform base + 32-bit lane offset (its issue phase 3,580 -> 1,390 cycles a step: it was the last to arrive), raised
//   priority for the producer and the dQ waves, the dQ stores off the KV waves (a KV wave's turn as storer cost ~2,400 cycles: four
//   serialized LDS-read -> store pairs right after the barrier), packed f32 math and one-instruction 16-bit packing in the KV waves,
//   transpose reads with immediate offsets (inline asm + counted lgkmcnt waits: the builtin takes none).  What did not: running the
//   dK / dV half of a step one step late in every second KV wave (-4 %, costs registers), 32 keys per KV wave (KT = 2: 7 KV waves,
//   10 waves, 168 registers, half the LDS reads and 44 % of the KV cycles per key: 204-209 us, also with the roles spread over the SIMDs
//   as {3 KV}, {2 KV + producer}, {KV + dQ}, {KV + dQ} in a 12-wave workgroup -- three of its KV waves on one SIMD take as long as
//   four of the small ones; the instantiation was removed again, the loops stay written over KT).
// ==========================================================================================================================
struct FusedArgs {
    TND q, k, v, dout, o;
    const float* lse;
    float* delta;            // NOT written (the two-kernel path's scratch): delta lives in LDS here
    OND dq, dk, dv;
    int H, N, nitems;
    float scale;
    int dbg;                 // experiments (VITK_ATTN_DBG): 1 = no DMA, 2 = KV waves skip their arithmetic, 4 = dQ waves skip theirs,
                             // 8 = cycle stamps of one wave per role, 16 = every wave's barrier wait (workgroup 0, into `delta`)
};
constexpr int FB_NKS = 7;                       // 32-row steps (192 < N <= 224 rows staged; key tiles limited to 13 -> N <= 208)
constexpr int FB_ROWS = FB_NKS * 32;            // 224
constexpr int FB_KROWS = 13 * 16;               // K / V rows staged (the key tiles): reads of rows 208..223 (the second half of the
                                                // 7th 32-key step) fall into the NEXT LDS region -- finite data -- and meet zero dS^T rows
constexpr int FB_TILE = FB_KROWS * 128;         // one staged K or V
constexpr int FB_DS = FB_ROWS * 64;             // dS^T of one step: 224 key rows x 32 queries (rows >= 16 nkt stay zero)
constexpr int FB_NST = 5;
// KT = 16-key tiles per KV wave: 1 -> 13 KV waves (16 waves, 128 registers a lane), the only flavour instantiated; 2 -> 7 KV waves (10
// waves, 168 registers; measured level, see the header: the code below stays generic in KT); + two dQ waves + the producer
constexpr int fb_kvw(int KT) { return KT == 1 ? 13 : 7; }
constexpr int fb_threads(int KT) { return (fb_kvw(KT) + 3) * 64; }
constexpr int FB_STAGE = 4096 + 4096 + 256;     // Q | dO | lse of a 32-query step
constexpr int FB_DQ = 32 * 128;                 // dQ of one step on its way from the dQ waves to the wave that stores it
constexpr int FB_OFF_K = 0, FB_OFF_V = 2 * FB_TILE, FB_OFF_DS = 3 * FB_TILE, FB_OFF_RING = FB_OFF_DS + 2 * FB_DS,
              FB_OFF_DQ = FB_OFF_RING + FB_NST * FB_STAGE, FB_OFF_DEL = FB_OFF_DQ + 2 * FB_DQ, FB_LDS = FB_OFF_DEL + 512;      // delta [2][32] | -log2(e) lse [2][32]
static_assert(FB_LDS <= 160 * 1024, "fused attention backward: LDS image");

// dS^T image: row = key, 64 bytes = 32 queries; 16-byte chunk c of row r sits at position c ^ (((r >> 2) & 1) << 1), which keeps
// the 8 rows x 32 bytes of a transpose read on 64 distinct banks
__device__ __forceinline__ int ds_off(int row, int qbyte) { return row * 64 + ((((qbyte >> 4) ^ (((row >> 2) & 1) << 1)) << 4) | (qbyte & 15)); }
// lane (i, g) gets dS^T[key0 + {4g..4g+3, 16+4g..16+4g+3}][q0 + i]  (q0 = 0 or 16)
__device__ __forceinline__ bf16x8 ds_tr(const char* buf, int key0, int q0, int fi, int fg) {
    const int row = key0 + 4 * fg + (fi >> 2);
    const char* p = buf + ds_off(row, (q0 + (fi & 3) * 4) * 2);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + 16 * 64));   // row + 16: same swizzle
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// two f32 -> one register of two 16-bit values (round to nearest even), one instruction
__device__ __forceinline__ unsigned fb_pk2(float lo, float hi) {
#ifdef VITK_HALF_IS_F16
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    const h2 r = {(_Float16)lo, (_Float16)hi}; return __builtin_bit_cast(unsigned, r);
#else
    unsigned r; asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi)); return r;
#endif
}
// cycle stamps (VITK_ATTN_DBG bit 3): ordered against the surrounding code, LDS results included
__device__ __forceinline__ unsigned long long fb_now() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
template <int KT>
__global__ __launch_bounds__(fb_threads(KT)) void attn_bwd_fused_kernel(const FusedArgs a) {
    constexpr int FB_KVW = fb_kvw(KT);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = a.N, H = a.H;
    const int first = blockIdx.x, stride = gridDim.x;
    const int nit = first < a.nitems ? (a.nitems - first + stride - 1) / stride : 0;      // items of this workgroup
    const int G = nit * FB_NKS;                                                           // steps of this workgroup
    if (nit == 0) return;
    char* const ring = smem + FB_OFF_RING;
    float* const del = reinterpret_cast<float*>(smem + FB_OFF_DEL);
    auto item_of = [&](int it) { return first + it * stride; };

    const int kvw = wave;                                          // KV waves first, then the two dQ waves, then the producer
    if (wave == FB_KVW + 2) {
        // ------------------------------------------------------------------ producer ------------------------------------------------
        const int lrow = lane >> 3, lchunk = (lane & 7) ^ (lane >> 3);
        // every piece: uniform item base (SGPR pair) + a 32-bit lane offset = min(first row + lrow, N - 1) * row stride + this lane's chunk
        __builtin_amdgcn_s_setprio(3);                             // the other waves wait for what this one issues
        const unsigned ch2 = lchunk * 16;
        const int sq = (int)a.q.s_n * 2, sd = (int)a.dout.s_n * 2, sk = (int)a.k.s_n * 2, sv = (int)a.v.s_n * 2;      // row strides, bytes
        auto piece = [&](const char* base, int row0, int stride, char* dst) {
            int row = row0 + lrow; row = row < N ? row : N - 1;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(base + (unsigned)(row * stride + ch2)),
                                             (void __attribute__((address_space(3)))*)dst, 16, 0, 0);
        };
        auto issue_stage = [&](int g) {              // 9 pieces: Q 4, dO 4, lse 1 (steps past the end re-load the last one: uniform counts)
            if (AP_DBG(a) & 1) return;
            const int gg = g < G ? g : G - 1;
            const int it = gg / FB_NKS, s = gg - it * FB_NKS;
            const int item = item_of(it), b = item / H, h = item - b * H;
            char* st = ring + (g % FB_NST) * FB_STAGE;
            const char* qb = reinterpret_cast<const char*>(a.q.p + b * a.q.s_b + h * a.q.s_h);
            const char* db = reinterpret_cast<const char*>(a.dout.p + b * a.dout.s_b + h * a.dout.s_h);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                piece(qb, s * 32 + 8 * j, sq, st + j * 1024);
                piece(db, s * 32 + 8 * j, sd, st + 4096 + j * 1024);
            }
            int qi = s * 32 + (lane & 31); qi = qi < N ? qi : N - 1;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(reinterpret_cast<const char*>(a.lse + (long long)item * N) + (unsigned)(qi * 4)),
                                             (void __attribute__((address_space(3)))*)(st + 8192), 4, 0, 0);
        };
        auto issue_kv = [&](int it, int part) {      // 13 pieces: part 0..3 of the 52 row groups of K (26) and V (26) of item `it` (clamped)
            if (AP_DBG(a) & 1) return;
            const int itc = it < nit ? it : nit - 1;
            const int item = item_of(itc), b = item / H, h = item - b * H;
            const char* kb = reinterpret_cast<const char*>(a.k.p + b * a.k.s_b + h * a.k.s_h);
            const char* vb = reinterpret_cast<const char*>(a.v.p + b * a.v.s_b + h * a.v.s_h);
            char* Kd = smem + FB_OFF_K + (it & 1) * FB_TILE;
            char* Vd = smem + FB_OFF_V;
            // 52 pieces = K groups 0..25 then V groups 0..25; part p issues pieces 13 p .. 13 p + 12
#pragma unroll
            for (int j = 0; j < 13; ++j) {
                const int pc = part * 13 + j;
                const bool isv = pc >= 26;
                const int g = isv ? pc - 26 : pc;
                piece(isv ? vb : kb, 8 * g, isv ? sv : sk, (isv ? Vd : Kd) + g * 1024);
            }
        };
        // prologue: K / V of the first item, ring stages 0..3; everything landed before the first barrier
        for (int part = 0; part < 4; ++part) issue_kv(0, part);
        for (int g = 0; g < 4; ++g) issue_stage(g);
        AP_WAIT_DMA();
        AP_BARRIER();          // P: stage 0 (and 1) visible -> the dQ waves form delta(0)
        AP_BARRIER();          // b_0
        const bool prof = (AP_DBG(a) & 24) && blockIdx.x == 0;
        unsigned long long tp[3] = {0, 0, 0}, t0 = 0, t1 = 0, t2 = 0;
        for (int g = 0; g <= G + 1; ++g) {
            const int s = g % FB_NKS, it = g / FB_NKS;
            if (prof) t0 = fb_now();
            issue_stage(g + 4);                                    // into the slot of step g - 1
            const bool kv = s >= 1 && s <= 4;
            if (kv) issue_kv(it + 1, s - 1);
            {   // dQ of step g - 2 (left in LDS by the dQ waves during step g - 1) out as whole 128-byte rows: exactly four store
                // instructions every step -- rows that do not exist (and steps 0, 1) go to a dump slot in the unused delta buffer
                const int gp = g >= 2 ? g - 2 : 0, itp = gp / FB_NKS, sp = gp - itp * FB_NKS;
                const int item = item_of(itp), b = item / H, h = item - b * H;
                const char* stg = smem + FB_OFF_DQ + ((g - 1) & 1) * FB_DQ;
                char* dqb = reinterpret_cast<char*>(reinterpret_cast<__bf16*>(a.dq.p) + b * a.dq.s_b + h * a.dq.s_h);
                char* dump = reinterpret_cast<char*>(a.delta) + (long long)blockIdx.x * N * 4 + 320 + (lane & 15) * 16;
                const int sdq = (int)a.dq.s_n * 2;
                bf16x8 v[4];
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4) v[j4] = *reinterpret_cast<const bf16x8*>(stg + (8 * j4 + lrow) * 128 + ((lane & 7) << 4));
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4) {
                    const int row = 8 * j4 + lrow, qi = sp * 32 + row;
                    char* dst = (g >= 2 && qi < N) ? dqb + (unsigned)(qi * sdq + (((lane & 7) ^ (row & 7)) << 4)) : dump;
                    *reinterpret_cast<bf16x8*>(dst) = v[j4];
                }
            }
            if (prof) t1 = fb_now();
            // before b_{g+1}: stage g + 2 and everything older has landed
            // (a step = 9 ring pieces + 13 K / V pieces in steps 1-4 + 4 stores, in this order)
            // stage g + 2 was issued in step g - 2: the operations of steps g - 1 and g may fly
            const bool kv_prev = s >= 2 && s <= 5;
            if (kv && kv_prev) asm volatile("s_waitcnt vmcnt(52)" ::: "memory");
            else if (kv || kv_prev) asm volatile("s_waitcnt vmcnt(39)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
            if (prof) t2 = fb_now();
            AP_BARRIER();                                          // b_{g+1}
            if (prof) { tp[0] += t1 - t0; tp[1] += t2 - t1; tp[2] += fb_now() - t2; }
        }
        if (prof && lane == 0) { unsigned long long* o = reinterpret_cast<unsigned long long*>(a.delta); o[16] = tp[0]; o[17] = tp[1]; o[18] = tp[2]; o[20 + wave] = tp[2]; }
        AP_WAIT_DMA();         // the surplus pieces must not outlive the workgroup's LDS
        return;
    }

    const int fi = lane & 15, fg = lane >> 4;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const float c = a.scale * LOG2E;

    if (wave >= FB_KVW) {
        // ------------------------------------------------------------------ dQ waves ------------------------------------------------
        const int qt = wave - FB_KVW;
        __builtin_amdgcn_s_setprio(2);                             // the longest per-step chain of the workgroup                                   // delta: which 16 of the step's 32 queries; dQ: which 32 columns of d
        struct ORow { bf16x8 o[2]; };
        auto o_rows = [&](int g) {                                  // the forward's output rows of step g (clamped), this lane's 16 columns
            const int gg = g < G ? g : G - 1;
            const int it = gg / FB_NKS, s = gg - it * FB_NKS;
            const int item = item_of(it), b = item / H, h = item - b * H;
            int qi = s * 32 + qt * 16 + fi; qi = qi < N ? qi : N - 1;
            const __bf16* op = a.o.p + b * a.o.s_b + h * a.o.s_h + (long long)qi * a.o.s_n;
            ORow r;
            r.o[0] = *reinterpret_cast<const bf16x8*>(op + 8 * fg);
            r.o[1] = *reinterpret_cast<const bf16x8*>(op + 32 + 8 * fg);
            return r;
        };
        auto make_delta = [&](int g, const ORow& r) {               // delta of step g from its ring stage (dO) and the O rows
            const char* dst = ring + (g % FB_NST) * FB_STAGE + 4096;
            float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 d8 = sw_row(dst, qt * 16, ks, fi, fg);
#pragma unroll
                for (int e = 0; e < 8; ++e) part[2 * ks + (e & 1)] = fmaf((float)d8[e], (float)r.o[ks][e], part[2 * ks + (e & 1)]);
            }
            float sum = groups_sum((part[0] + part[1]) + (part[2] + part[3]));
            if (fg == 0) {
                del[(g & 1) * 32 + qt * 16 + fi] = -sum;            // negated: the KV waves start dP's accumulator from it
                // the exp2 argument's constant term for the KV waves (one FMA there instead of a multiply and an FMA)
                del[64 + (g & 1) * 32 + qt * 16 + fi] = -LOG2E * reinterpret_cast<const float*>(dst + 4096)[qt * 16 + fi];
            }
        };
        ORow cur = o_rows(0);
        ORow nxt = o_rows(1);
        AP_BARRIER();                                              // P
        make_delta(0, cur);
        cur = nxt;                                                 // rows of step 1
        nxt = o_rows(2);
        AP_BARRIER();                                              // b_0
        constexpr int KREG = 5;                                    // K^T key steps held in registers per item (the same fragments serve all 7 query
                                                                   // steps); the rest is re-read from LDS each step: 128 registers a lane
        bf16x8 ktf[KREG][2];
        const bool prof = blockIdx.x == 0 && (((AP_DBG(a) & 8) && qt == 0) || (AP_DBG(a) & 16));
        unsigned long long tp[3] = {0, 0, 0}, t0 = 0, t1 = 0, t2 = 0;
        for (int g = 0; g <= G + 1; ++g) {
            if (prof) t0 = fb_now();
            const ORow req = o_rows(g + 3);                        // requested now, used two steps from now
            if (g >= 1 && g <= G && !(AP_DBG(a) & 4)) {                // dQ of step g - 1: dS K over all keys, columns 32 qt .. 32 qt + 31 of d
                const int gp = g - 1, itp = gp / FB_NKS;
                const char* dsb = smem + FB_OFF_DS + (gp & 1) * FB_DS;
                const char* Ks = smem + FB_OFF_K + (itp & 1) * FB_TILE;
                if (gp - itp * FB_NKS == 0) {
#pragma unroll
                    for (int ks = 0; ks < KREG; ++ks)
#pragma unroll
                        for (int f = 0; f < 2; ++f) ktf[ks][f] = sw_tr(Ks, ks * 32, (2 * qt + f) * 16, fi, fg);
                }
                f32x4 acc[2][2];
                acc[0][0] = acc[0][1] = acc[1][0] = acc[1][1] = z4;
#pragma unroll
                for (int ks = 0; ks < FB_NKS; ++ks) {
                    const bf16x8 ds0 = ds_tr(dsb, ks * 32, 0, fi, fg), ds1 = ds_tr(dsb, ks * 32, 16, fi, fg);
#pragma unroll
                    for (int f = 0; f < 2; ++f) {
                        const bf16x8 kt = ks < KREG ? ktf[ks][f] : sw_tr(Ks, ks * 32, (2 * qt + f) * 16, fi, fg);
                        acc[0][f] = MFMA(kt, ds0, acc[0][f]);
                        acc[1][f] = MFMA(kt, ds1, acc[1][f]);
                    }
                }
                // -> LDS (128-byte rows, chunk c of row r at position c ^ (r & 7)): lane (i, g) holds 4 consecutive d of query 16 t + i
                char* stg = smem + FB_OFF_DQ + (g & 1) * FB_DQ;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int f = 0; f < 2; ++f) {
                        const int row = t * 16 + fi, chunk = 2 * (2 * qt + f) + (fg >> 1);
                        const f32x4 v = acc[t][f] * a.scale;
                        const bf16x4 o4 = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                        *reinterpret_cast<bf16x4*>(stg + row * 128 + ((chunk ^ (row & 7)) << 4) + ((fg & 1) << 3)) = o4;
                    }
            }
            if (prof) t1 = fb_now();
            if (g + 1 < G) make_delta(g + 1, cur);                 // for the step ahead: its ring stage landed before b_g
            cur = nxt; nxt = req;
            if (prof) t2 = fb_now();
            AP_BARRIER();                                          // b_{g+1}
            if (prof) { tp[0] += t1 - t0; tp[1] += t2 - t1; tp[2] += fb_now() - t2; }
        }
        if (prof && lane == 0) {
            unsigned long long* o = reinterpret_cast<unsigned long long*>(a.delta);
            if (AP_DBG(a) & 16) o[20 + wave] = tp[2];
            if (qt == 0) { o[8] = tp[0]; o[9] = tp[1]; o[10] = tp[2]; }
        }
        return;
    }

    // ---------------------------------------------------------------------- KV waves ------------------------------------------------
    // wave w: keys 16 KT w .. 16 KT w + 16 KT - 1 as KT 16-key tiles (KT = 2: every Q / dO fragment read from LDS feeds two MFMAs)
    const int nkt = (N + 15) >> 4;
    int ki[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) ki[t] = (kvw * KT + t) * 16 + fi;
    const bool pad_wave = (kvw + 1) * KT * 16 > N;                 // holds padding keys (wave-uniform)
    const int kvt = kvw * 64 + lane;                               // thread index among the KV waves
    // rows 16 nkt .. 223 of both dS^T buffers: zero them once (the dQ waves sum over all 224 rows)
    for (int i = kvt; i < 2 * FB_DS / 16; i += FB_KVW * 64) {
        const int bufi = i / (FB_DS / 16), r = (i % (FB_DS / 16)) >> 2;
        if (r >= nkt * 16) *reinterpret_cast<f32x4*>(smem + FB_OFF_DS + bufi * FB_DS + (i % (FB_DS / 16)) * 16) = z4;
    }
    // K[1] is empty during the first item and reads of K[0] rows 208..223 fall into its first 16 rows: finite data there
    if (kvt < 128) *reinterpret_cast<f32x4*>(smem + FB_OFF_K + FB_TILE + kvt * 16) = z4;
    AP_BARRIER();                                                  // P
    AP_BARRIER();                                                  // b_0
    bf16x8 kf[KT][2], vf[KT][2];
    f32x4 accK[KT][4], accV[KT][4];
    bf16x8 pb[KT], dsb[KT];                                        // P and dS of the step, 16-bit: from the first half to the second
    // lane offsets into a ring stage (the 128-byte-row swizzle of sw_row / sw_tr, rows 0..15): one VGPR add per step each, everything
    // else of an access is an immediate (query half + 2048, dO + 4096)
    unsigned lrow_off[2], ltr_off[4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) lrow_off[ks] = fi * 128 + (((ks * 4 + fg) ^ (fi & 7)) << 4);
#pragma unroll
    for (int fd = 0; fd < 4; ++fd) {
        const int row = 4 * fg + (fi >> 2);
        ltr_off[fd] = row * 128 + ((((fd * 2) + ((fi & 3) >> 1)) ^ (row & 7)) << 4) + ((fi & 1) << 3);
    }
    const unsigned smem_lo = (unsigned)(size_t)smem;               // LDS address of the image (the low half of the flat address)
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    const f32x2 c2 = {c, c};

    // first half of step g: S, dP, P, dS for this wave's keys x the step's 32 queries; dS^T -> LDS for the dQ waves
    auto front = [&](int g) {
        const int it = g / FB_NKS, s = g - it * FB_NKS;
        if (s == 0) {
            const char* Ks = smem + FB_OFF_K + (it & 1) * FB_TILE;
            const char* Vs = smem + FB_OFF_V;
            // (KT = 2: the last wave's second tile reads rows 208..223: past the staged tile, into the next LDS region -- its products are masked)
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                const int r0 = (kvw * KT + t) * 16;
                kf[t][0] = sw_row(Ks, r0, 0, fi, fg); kf[t][1] = sw_row(Ks, r0, 1, fi, fg);
                vf[t][0] = sw_row(Vs, r0, 0, fi, fg); vf[t][1] = sw_row(Vs, r0, 1, fi, fg);
            }
        }
        const unsigned stage = FB_OFF_RING + (g % FB_NST) * FB_STAGE;
        const char* rb0 = smem + (stage + lrow_off[0]);        // Q rows, d 0..31 (this lane's 16 bytes); + 2048: queries 16..31; + 4096: dO
        const char* rb1 = smem + (stage + lrow_off[1]);        // d 32..63
        const float* del_s = del + (g & 1) * 32;               // -delta
        const float* nl_s = del_s + 64;                        // -log2(e) lse, written by the dQ waves with delta
        u32x2 ph[KT][2], dh[KT][2];                            // [tile][query half]: this lane's 4 queries x 1 key, 16-bit
        const bool last_step = s == FB_NKS - 1;                // padding query rows only exist in the last step
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int row0 = hh * 16;
            const bf16x8 q0 = *reinterpret_cast<const bf16x8*>(rb0 + hh * 2048), q1 = *reinterpret_cast<const bf16x8*>(rb1 + hh * 2048);
            const bf16x8 d0 = *reinterpret_cast<const bf16x8*>(rb0 + 4096 + hh * 2048), d1 = *reinterpret_cast<const bf16x8*>(rb1 + 4096 + hh * 2048);
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(nl_s + row0 + 4 * fg);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(del_s + row0 + 4 * fg);
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                f32x4 st = MFMA(q0, kf[t][0], z4);             // S[q = row0 + 4g + e][key]
                st = MFMA(q1, kf[t][1], st);
                f32x4 dp = MFMA(d0, vf[t][0], d4);             // dP[q][key] - delta[q]
                dp = MFMA(d1, vf[t][1], dp);
                f32x4 pv, dv;
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) {               // float pairs: v_pk_fma_f32 / v_pk_mul_f32
                    const f32x2 x = __builtin_elementwise_fma(f32x2{st[2 * e2], st[2 * e2 + 1]}, c2, f32x2{l4[2 * e2], l4[2 * e2 + 1]});
                    pv[2 * e2] = __builtin_amdgcn_exp2f(x[0]); pv[2 * e2 + 1] = __builtin_amdgcn_exp2f(x[1]);
                    const f32x2 y = f32x2{pv[2 * e2], pv[2 * e2 + 1]} * f32x2{dp[2 * e2], dp[2 * e2 + 1]};      // `scale` is applied once, to dK and dQ
                    dv[2 * e2] = y[0]; dv[2 * e2 + 1] = y[1];
                }
                if (last_step) {                               // (the empty asm keeps these wave-uniform tests branches: 24 selects a step otherwise)
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (s * 32 + row0 + 4 * fg + e >= N) { pv[e] = 0.f; dv[e] = 0.f; }
                }
                if (pad_wave) {                                // padding keys: nothing for dQ; their dK / dV rows are not stored
                    asm volatile("" ::: "memory");
                    if (ki[t] >= N) { pv = z4; dv = z4; }
                }
                ph[t][hh] = u32x2{fb_pk2(pv[0], pv[1]), fb_pk2(pv[2], pv[3])};
                dh[t][hh] = u32x2{fb_pk2(dv[0], dv[1]), fb_pk2(dv[2], dv[3])};
            }
        }
        char* dsw = smem + FB_OFF_DS + (g & 1) * FB_DS;
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            pb[t] = __builtin_bit_cast(bf16x8, u32x4{ph[t][0][0], ph[t][0][1], ph[t][1][0], ph[t][1][1]});
            dsb[t] = __builtin_bit_cast(bf16x8, u32x4{dh[t][0][0], dh[t][0][1], dh[t][1][0], dh[t][1][1]});
            // dS^T -> LDS: this lane's key row, queries {4g..4g+3} and {16+4g..+3} of the step
            *reinterpret_cast<u32x2*>(dsw + ds_off(ki[t], 8 * fg)) = dh[t][0];
            *reinterpret_cast<u32x2*>(dsw + ds_off(ki[t], 32 + 8 * fg)) = dh[t][1];
        }
    };
    // second half of step g: dV^T += dO^T P, dK^T += Q^T dS from the step's ring stage (transpose reads); the item's rows out after its last step
    auto back = [&](int g) {
        const int it = g / FB_NKS, s = g - it * FB_NKS;
        const unsigned stage = FB_OFF_RING + (g % FB_NST) * FB_STAGE;
        if (s == 0) {
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int fd = 0; fd < 4; ++fd) { accK[t][fd] = z4; accV[t][fd] = z4; }
        }
        // The 16 transpose reads as inline asm: the builtin takes no immediate offset (one VALU add per read); here one add per 16-column
        // block.  The compiler does not track them: each block's four results are released by a counted lgkmcnt wait they pass through
        // (LDS operations return in order; anything the compiler issues in between only makes the counts conservative).
        constexpr int NB = KT == 1 ? 4 : 2;                    // 16-column blocks read ahead at a time (registers: 8 a block)
#pragma unroll
        for (int f0 = 0; f0 < 4; f0 += NB) {
            u32x2 r[NB][4];                                    // [block][dO lo, dO hi, Q lo, Q hi]: rows {4g..4g+3} / {16+4g..+3}, transposed
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const unsigned tb = smem_lo + stage + ltr_off[f0 + j];
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(r[j][0]) : "v"(tb));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:6144" : "=v"(r[j][1]) : "v"(tb));
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r[j][2]) : "v"(tb));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(r[j][3]) : "v"(tb));
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (j) __builtin_amdgcn_sched_barrier(0);      // keep each wait behind the previous block's MFMAs
                const int left = 4 * (NB - 1 - j);
                if (left == 12) asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(r[j][0]), "+v"(r[j][1]), "+v"(r[j][2]), "+v"(r[j][3]));
                if (left == 8) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(r[j][0]), "+v"(r[j][1]), "+v"(r[j][2]), "+v"(r[j][3]));
                if (left == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(r[j][0]), "+v"(r[j][1]), "+v"(r[j][2]), "+v"(r[j][3]));
                if (left == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[j][0]), "+v"(r[j][1]), "+v"(r[j][2]), "+v"(r[j][3]));
                const bf16x8 dot = __builtin_bit_cast(bf16x8, u32x4{r[j][0][0], r[j][0][1], r[j][1][0], r[j][1][1]});
                const bf16x8 qt8 = __builtin_bit_cast(bf16x8, u32x4{r[j][2][0], r[j][2][1], r[j][3][0], r[j][3][1]});
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    accV[t][f0 + j] = MFMA(dot, pb[t], accV[t][f0 + j]);      // dV^T[d][key]
                    accK[t][f0 + j] = MFMA(qt8, dsb[t], accK[t][f0 + j]);     // dK^T[d][key]
                }
            }
        }
        if (s == FB_NKS - 1) {
            // the item's dK / dV rows: uniform item base + a 32-bit lane offset formed here (nothing of it lives through the steps)
            const int item = item_of(it), b = item / H, h = item - b * H;
            char* dkb = reinterpret_cast<char*>(reinterpret_cast<__bf16*>(a.dk.p) + b * a.dk.s_b + h * a.dk.s_h);
            char* dvb = reinterpret_cast<char*>(reinterpret_cast<__bf16*>(a.dv.p) + b * a.dv.s_b + h * a.dv.s_h);
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));                   // opaque: keeps the offsets below out of the loop-invariant (spilled) set
            const int fi_o = lane_o & 15, fg_o = lane_o >> 4;
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                const int key = (kvw * KT + t) * 16 + fi_o;
                // [round 6] lane pairs (fg, fg ^ 1) exchange halves (v_permlane16_swap) so that a lane stores two 16-byte pieces -- 8 consecutive
                // columns of two 16-column blocks -- per tensor instead of four 8-byte ones (as in attention.hip's forward)
                typedef unsigned fb_u4 __attribute__((ext_vector_type(4)));
                const fb_u4 kx = __builtin_bit_cast(fb_u4, pack8(accK[t][0] * a.scale, accK[t][1] * a.scale)), ky = __builtin_bit_cast(fb_u4, pack8(accK[t][2] * a.scale, accK[t][3] * a.scale));
                const fb_u4 vx = __builtin_bit_cast(fb_u4, pack8(accV[t][0], accV[t][1])), vy = __builtin_bit_cast(fb_u4, pack8(accV[t][2], accV[t][3]));
                unsigned ka[4], kb[4], va[4], vb[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    auto sk = __builtin_amdgcn_permlane16_swap(kx[d], ky[d], false, false);
                    auto sv = __builtin_amdgcn_permlane16_swap(vx[d], vy[d], false, false);
                    ka[d] = (unsigned)sk[0]; kb[d] = (unsigned)sk[1]; va[d] = (unsigned)sv[0]; vb[d] = (unsigned)sv[1];
                }
                if (key < N) {
                    const unsigned col = ((fg_o & 1) ? 64u : 0u) + 8u * (fg_o & 2);          // bytes: block 2 / 0, + 8 columns for fg 2, 3
                    char* dkp = dkb + (unsigned)(key * ((int)a.dk.s_n * 2)) + col;
                    char* dvp = dvb + (unsigned)(key * ((int)a.dv.s_n * 2)) + col;
                    // [measured: a second exchange level that makes each instruction's pieces of a row contiguous (64 bytes) is level -- it is
                    //  the NUMBER of store instructions queued behind the producer's DMA stream that counted (profiles/r06ab_store_widening_ab.log)]
                    *reinterpret_cast<fb_u4*>(dkp) = fb_u4{ka[0], ka[1], kb[0], kb[1]};
                    *reinterpret_cast<fb_u4*>(dkp + 32) = fb_u4{ka[2], ka[3], kb[2], kb[3]};
                    *reinterpret_cast<fb_u4*>(dvp) = fb_u4{va[0], va[1], vb[0], vb[1]};
                    *reinterpret_cast<fb_u4*>(dvp + 32) = fb_u4{va[2], va[3], vb[2], vb[3]};
                }
            }
        }
    };

    const bool work = !(AP_DBG(a) & 2);
    const bool prof = blockIdx.x == 0 && (((AP_DBG(a) & 8) && wave == 2) || (AP_DBG(a) & 16));
    unsigned long long tp[4] = {0, 0, 0, 0}, t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    for (int g = 0; g <= G + 1; ++g) {
        if (prof) t0 = fb_now();
        if (prof) t1 = fb_now();
        if (work) {
            if (g < G) front(g);
            if (prof) t2 = fb_now();
            if (g < G) back(g);
        }
        if (prof) t3 = fb_now();
        AP_BARRIER();                                              // b_{g+1}
        if (prof) { tp[0] += t1 - t0; tp[1] += t2 - t1; tp[2] += t3 - t2; tp[3] += fb_now() - t3; }
    }
    if (prof && lane == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(a.delta);
        if (AP_DBG(a) & 16) o[20 + wave] = tp[3];
        if (wave == 2) { o[0] = tp[0]; o[1] = tp[1]; o[2] = tp[2]; o[3] = tp[3]; o[4] = (unsigned long long)G; }
    }
}

// ---- elementwise split of an f32 tensor into hi + lo 16-bit terms ----
__global__ __launch_bounds__(256) void split2_kernel(const float* __restrict__ x, __bf16* __restrict__ hi, __bf16* __restrict__ lo, long long n4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * i);
        bf16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) { h[e] = (__bf16)v[e]; l[e] = (__bf16)(v[e] - (float)h[e]); }
        *reinterpret_cast<bf16x4*>(hi + 4 * i) = h;
        *reinterpret_cast<bf16x4*>(lo + 4 * i) = l;
    }
}

int num_cus() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) return 256;
        return v;
    }();
    return n;
}
template <typename K> int set_lds(K kernel, int bytes) {
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
constexpr int AP_MAX_LDS = 160 * 1024;      // the fused backward's image is 153.5 KB; the others ask for at most 119 KB
#define AP_SET_LDS(kernel, name) do { static const int rc__ = set_lds(kernel, AP_MAX_LDS); \
    if (rc__ != 0) VITK_FAIL(rc__, "%s: hipFuncSetAttribute(max dynamic LDS) failed: %d", name, rc__); } while (0)

TND tnd(const vitk_bhnd& t) { return TND{(const __bf16*)t.p, (long long)t.s_b, (long long)t.s_h, (long long)t.s_n}; }
OND ond(const vitk_bhnd& t) { return OND{t.p, (long long)t.s_b, (long long)t.s_h, (long long)t.s_n}; }

template <int NS> int launch_fwd(const AttnPipeFwd& p, hipStream_t st) {
    FwdArgs<NS> a;
    for (int t = 0; t < NS; ++t) { a.q[t] = tnd(p.q[t]); a.k[t] = tnd(p.k[t]); a.v[t] = tnd(p.v[t]); }
    a.o = ond(p.o); a.lse = p.lse; a.H = (int)p.H; a.N = (int)p.N; a.nitems = (int)(p.B * p.H);
    a.dbg = vitk_exp("VITK_ATTN_DBG") ? atoi(vitk_exp("VITK_ATTN_DBG")) : 0;
    a.c = p.scale * LOG2E; a.drop_t = drop_thresh(p.drop_p); a.drop_seed = p.drop_seed; a.inv_keep = 1.0f / (1.0f - p.drop_p);
    const int nks = (int)((p.N + 31) / 32);
    const int lds = NS * 2 * nks * 32 * 128;
    const int per_cu = NS == 1 ? 2 : 1;
    int grid = per_cu * num_cus();
    if (grid > a.nitems) grid = a.nitems;
    if (NS == 1 && a.drop_t) {
        AP_SET_LDS((attn_fwd_pipe_kernel<1, true>), "attn_fwd (pipelined)");
        FwdArgs<1> a1; memcpy(&a1, &a, sizeof(a1) < sizeof(a) ? sizeof(a1) : sizeof(a));
        hipLaunchKernelGGL((attn_fwd_pipe_kernel<1, true>), dim3((unsigned)grid), dim3(512), (size_t)lds, st, a1);
    } else {
        AP_SET_LDS((attn_fwd_pipe_kernel<NS, false>), "attn_fwd (pipelined)");
        hipLaunchKernelGGL((attn_fwd_pipe_kernel<NS, false>), dim3((unsigned)grid), dim3(512), (size_t)lds, st, a);
    }
    VITK_CHECK_LAUNCH("attn_fwd (pipelined)");
    return 0;
}
// which: bit 0 = the dQ kernel (also writes delta), bit 1 = the dK / dV kernel (reads delta)
template <int NS> int launch_bwd(const AttnPipeBwd& p, hipStream_t st, int which) {
    const int nks = (int)((p.N + 31) / 32);
    const int nbuf = NS == 1 ? 2 : 1;
    const int nitems = (int)(p.B * p.H);
    int grid = num_cus();
    if (grid > nitems) grid = nitems;
    const unsigned drop_t = drop_thresh(p.drop_p);
    const float inv_keep = 1.0f / (1.0f - p.drop_p);
    const int dbg = vitk_exp("VITK_ATTN_DBG") ? atoi(vitk_exp("VITK_ATTN_DBG")) : 0;
    if (which & 1) {
        DqArgs<NS> a;
        a.dbg = dbg;
        for (int t = 0; t < NS; ++t) { a.q[t] = tnd(p.q[t]); a.k[t] = tnd(p.k[t]); a.v[t] = tnd(p.v[t]); a.dout[t] = tnd(p.dout[t]); }
        a.o = ond(p.o); a.lse = p.lse; a.delta = p.delta; a.dq = ond(p.dq);
        a.H = (int)p.H; a.N = (int)p.N; a.nitems = nitems; a.scale = p.scale; a.drop_t = drop_t; a.drop_seed = p.drop_seed; a.inv_keep = inv_keep;
        const size_t lds = (size_t)(nbuf * NS * 2 * nks * 32 * 128);
        if (NS == 1 && drop_t) {
            AP_SET_LDS((attn_dq_pipe_kernel<1, true>), "attn_bwd_dq (pipelined)");
            DqArgs<1> a1; memcpy(&a1, &a, sizeof(a1) < sizeof(a) ? sizeof(a1) : sizeof(a));
            hipLaunchKernelGGL((attn_dq_pipe_kernel<1, true>), dim3((unsigned)grid), dim3(1024), lds, st, a1);
        } else {
            AP_SET_LDS((attn_dq_pipe_kernel<NS, false>), "attn_bwd_dq (pipelined)");
            hipLaunchKernelGGL((attn_dq_pipe_kernel<NS, false>), dim3((unsigned)grid), dim3(1024), lds, st, a);
        }
        VITK_CHECK_LAUNCH("attn_bwd_dq (pipelined)");
    }
    if (which & 2) {
        DkvArgs<NS> a;
        a.dbg = dbg;
        for (int t = 0; t < NS; ++t) { a.q[t] = tnd(p.q[t]); a.k[t] = tnd(p.k[t]); a.v[t] = tnd(p.v[t]); a.dout[t] = tnd(p.dout[t]); }
        a.lse = p.lse; a.delta = p.delta; a.dk = ond(p.dk); a.dv = ond(p.dv);
        a.H = (int)p.H; a.N = (int)p.N; a.nitems = nitems; a.scale = p.scale; a.drop_t = drop_t; a.drop_seed = p.drop_seed; a.inv_keep = inv_keep;
        const size_t lds = (size_t)(nbuf * (NS * 2 * nks * 32 * 128 + 2048));
        if (NS == 1 && drop_t) {
            AP_SET_LDS((attn_dkv_pipe_kernel<1, true>), "attn_bwd_dkv (pipelined)");
            DkvArgs<1> a1; memcpy(&a1, &a, sizeof(a1) < sizeof(a) ? sizeof(a1) : sizeof(a));
            hipLaunchKernelGGL((attn_dkv_pipe_kernel<1, true>), dim3((unsigned)grid), dim3(1024), lds, st, a1);
        } else {
            AP_SET_LDS((attn_dkv_pipe_kernel<NS, false>), "attn_bwd_dkv (pipelined)");
            hipLaunchKernelGGL((attn_dkv_pipe_kernel<NS, false>), dim3((unsigned)grid), dim3(1024), lds, st, a);
        }
        VITK_CHECK_LAUNCH("attn_bwd_dkv (pipelined)");
    }
    return 0;
}

int launch_fused(const AttnPipeBwd& p, hipStream_t st) {
    FusedArgs a;
    a.q = tnd(p.q[0]); a.k = tnd(p.k[0]); a.v = tnd(p.v[0]); a.dout = tnd(p.dout[0]); a.o = tnd(p.o);
    a.lse = p.lse; a.delta = p.delta; a.dq = ond(p.dq); a.dk = ond(p.dk); a.dv = ond(p.dv);
    a.H = (int)p.H; a.N = (int)p.N; a.nitems = (int)(p.B * p.H); a.scale = p.scale;
    a.dbg = vitk_exp("VITK_ATTN_DBG") ? atoi(vitk_exp("VITK_ATTN_DBG")) : 0;
    int grid = num_cus();
    if (grid > a.nitems) grid = a.nitems;
    AP_SET_LDS(attn_bwd_fused_kernel<1>, "attn_bwd (fused)");
    hipLaunchKernelGGL(attn_bwd_fused_kernel<1>, dim3((unsigned)grid), dim3(fb_threads(1)), (size_t)FB_LDS, st, a);
    VITK_CHECK_LAUNCH("attn_bwd (fused)");
    return 0;
}

}  // namespace

bool attn_pipe_supported(int64_t N, int64_t d) { return d == 64 && N > 32 && N <= 224; }
bool attn_fused_bwd_supported(int64_t N, int64_t d, float drop_p) { return d == 64 && N > 192 && N <= 208 && drop_p == 0.f; }
int attn_pipe_bwd_fused(const AttnPipeBwd& a, void* stream) { return launch_fused(a, (hipStream_t)stream); }
// Which 16-bit kernels run pipelined by default: bit 0 forward, bit 1 dQ, bit 2 dK/dV.  [measured, ViT-B/16 batch 256, rocprofv3
// averages, pipelined vs one-workgroup-per-head] forward 89 vs 81 us, dQ 110 vs 118 us, dK/dV 150 vs 143 us: only the dQ kernel -- the
// one whose per-wave rows are prefetched a whole item ahead -- gains, so only it is on.  VITK_ATTN_PIPE=<mask> overrides (tests: 7).
int attn_pipe_mask() {
    const char* e = vitk_switch("VITK_ATTN_PIPE");
    if (e) return atoi(e);
    // the pipelined kernels are resident workgroups with STATIC item lists and a CU each: while another kernel is expected on the chip
    // (vitk_set_cu_reserve > 0: a collective overlapping the backward) the per-head kernels run instead -- their 3,072 independent
    // workgroups simply use the CUs that are there
    // default: the single-kernel backward where it applies (16-bit, 192 < N <= 208, no dropout: 216 us against 280 us for the
    // pair at ViT-B/16 batch 256), else the pipelined dQ kernel + the per-head dK / dV kernel; per-head forward
    return vitk_get_cu_reserve() > 0 ? 0 : (2 | 8);
}
int attn_pipe_fwd(const AttnPipeFwd& a, void* stream) {
    return a.ns == 2 ? launch_fwd<2>(a, (hipStream_t)stream) : launch_fwd<1>(a, (hipStream_t)stream);
}
int attn_pipe_bwd(const AttnPipeBwd& a, void* stream, int which) {
    return a.ns == 2 ? launch_bwd<2>(a, (hipStream_t)stream, which) : launch_bwd<1>(a, (hipStream_t)stream, which);
}

extern "C" int vitk_split2(const float* x, void* hi, void* lo, int64_t n, void* stream) {
    if (!x || !hi || !lo) VITK_FAIL(VITK_E_ARG, "split2: null pointer");
    if (n <= 0 || (n & 3)) VITK_FAIL(VITK_E_SHAPE, "split2: element count must be a positive multiple of 4 (got %lld)", (long long)n);
    if (!aligned16(x) || !aligned8(hi) || !aligned8(lo)) VITK_FAIL(VITK_E_ALIGN, "split2: x must be 16-byte, hi / lo 8-byte aligned");
    long long blocks = (n / 4 + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(split2_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (__bf16*)hi, (__bf16*)lo, (long long)(n / 4));
    VITK_CHECK_LAUNCH("split2");
    return 0;
}
