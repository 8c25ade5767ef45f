// attention.hip -- fused scaled-dot-product attention forward/backward for gfx950, bf16, d_head = 64.
//
// Replaces vit.py:55-63 (simple_vit.py:54-61): the three head-split rearranges, matmul(q,k^T)*scale,
// nn.Softmax, matmul(attn, v) and the head-merge rearrange -- and their autograd.  The N x N score
// matrix is never materialised; q/k/v are read in place from the merged (B, N, 3*h*d) to_qkv output
// and O / dQ,dK,dV are written in merged layouts, so no contiguity copies exist.
//
// ViT sequences are short (N = 197 for ViT-B/L at 224^2), so ONE 8-wave workgroup owns one (batch, head):
// the whole K and V of that head (2 x N x 128 B) are staged once into LDS and every wave carries R (1 or 2)
// 16-row query tiles against them (backward: query tiles in the dQ kernel, key tiles in the dK/dV kernel).
//
// Fragment algebra (v_mfma_f32_16x16x32_bf16; D[i][j] = sum_k A[i][k] B[k][j]; a lane holds
// D[4*(lane>>4)+r][lane&15], r = 0..3):
//   forward   S^T = K Q^T      A = K rows (ds_read_b128),           B = Q rows (registers)
//             O^T = V^T P^T    A = V^T  (ds_read_b64_tr_b16),       B = P^T straight from the S^T
//                              accumulators: the 4 keys a lane holds per fragment are exactly a legal
//                              k-slice of the B operand once V is read with the matching key order
//                              (keys {4g..4g+3} of two consecutive 16-key fragments), so P never
//                              leaves registers and no cross-lane transpose is needed.
//   bwd dQ    S^T, dP^T = V dO^T, dS^T = P^T*(dP^T - delta)*scale,  dQ^T = K^T dS^T   (query-tile outer)
//   bwd dK/dV S = Q K^T, dP = dO V^T (A from LDS rows of Q / dO, B = K / V rows in registers),
//             dV^T = dO^T P, dK^T = Q^T dS (A via transpose reads of dO / Q)          (key-tile outer)
// Softmax is online in the exp2 domain with a LAZY reference maximum and MFMA row sums (see attn_fwd_kernel);
// backward recomputes P from the saved row log-sum-exp.  LDS rows are padded to 160 B: conflict-free for both the b128 row
// reads and the b64 transpose reads.  The variable-length (packed) kernels -- NaViT, and every fixed-length shape these kernels
// do not take -- live in attention_varlen.hip; the rest of this file holds the head-wise q/k normalisation of NaViT and the
// materialising softmax pair of the fallback path.
#include "common.h"
#include "attention_frag.h"
#include "attention_pipe.h"

namespace {

constexpr int AT_LD = 160;  // bytes per LDS row: 64 bf16 + 32 B pad
constexpr int AT_THREADS = 512;  // 8 waves per (batch, head): query / key tiles are dealt round-robin to the waves
constexpr int AT_WAVES = AT_THREADS / 64;

struct BHND { __bf16* p; long long s_b, s_h, s_n; };
typedef unsigned q4u __attribute__((ext_vector_type(4)));

// Two tiles at once with every global load of a batch in flight before the first LDS store: a plain
// load -> store loop is serialised by the compiler (s_waitcnt vmcnt(0) per 16 bytes), which at ~2 us of
// HBM latency per trip was most of a (batch, head) workgroup's life.
// The staged rows are read once per workgroup: nontemporal loads (the same hint took the LayerNorm backward from 123 to 99 us;
// here: forward 81 -> 77.5 us, backward 276 -> 270 us).  On the per-tile row loads (q, dO, O / k, v in registers) it costs ~2 %.
__device__ __forceinline__ bf16x8 ld_nt8(const __bf16* p) { return __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p)); }
__device__ __forceinline__ void fill_tiles2(char* t0, const __bf16* s0, long long n0, char* t1, const __bf16* s1, long long n1,
                                            int N, int rows_pad, int tid) {
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const int lim = rows_pad * 8;
    for (int c0 = tid; c0 < lim; c0 += 4 * AT_THREADS) {
        bf16x8 va[4], vb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + j * AT_THREADS, row = c >> 3, col8 = c & 7;
            va[j] = zero8; vb[j] = zero8;
            if (c < lim && row < N) {
                va[j] = ld_nt8(s0 + (long long)row * n0 + col8 * 8);
                vb[j] = ld_nt8(s1 + (long long)row * n1 + col8 * 8);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + j * AT_THREADS, row = c >> 3, col8 = c & 7;
            if (c < lim) {
                *reinterpret_cast<bf16x8*>(t0 + row * AT_LD + col8 * 16) = va[j];
                *reinterpret_cast<bf16x8*>(t1 + row * AT_LD + col8 * 16) = vb[j];
            }
        }
    }
}
// 16 rows x (32 of the 64 columns) as an MFMA A/B operand: lane (i = lane&15, g = lane>>4) holds
// tile[row0 + i][ks*32 + 8g .. +7]
__device__ __forceinline__ bf16x8 row_frag(const char* tile, int row0, int ks, int fi, int fg) {
    return *reinterpret_cast<const bf16x8*>(tile + (row0 + fi) * AT_LD + (ks * 32 + 8 * fg) * 2);
}
// transposed operand: lane (i, g) holds tile[row0 + {4g..4g+3, 16+4g..16+4g+3}][col0 + i]
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int row0, int col0, int fi, int fg) {
    const char* p = tile + (row0 + 4 * fg + (fi >> 2)) * AT_LD + (col0 + (fi & 3) * 4) * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + 16 * AT_LD));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
// R = 16-row query tiles a wave carries at once: every K / V fragment read from LDS feeds R MFMAs, so R = 2 halves
// the LDS traffic per flop and gives the scheduler two independent softmax chains; with 8 waves it also covers
// N <= 256 (ViT-B/L: 13 tiles) in ONE pass instead of two unbalanced ones.
// DROP: attention dropout compiled in (round 6: a template parameter -- the runtime test cost two VALU instructions and a branch per tile
// and key step in the instance every inference / p = 0 training step runs)
template <int R, bool DROP = true>
__global__ __launch_bounds__(AT_THREADS) void attn_fwd_kernel(BHND q, BHND k, BHND v, BHND o, float* __restrict__ lse,
                                                        int H, int N, float scale_log2e, unsigned drop_t, unsigned drop_seed,
                                                        float inv_keep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, fg = lane >> 4;
    const int rows_pad = ((N + 31) >> 5) << 5;
    char* Ks = smem;
    char* Vs = Ks + rows_pad * AT_LD;
    const int nqt = (N + 15) >> 4, nks = rows_pad >> 5;
    int bh = 0, b = 0, h = 0, t0 = 0;
    const __bf16* qbase = q.p;
    bf16x8 qf[R][2];
    auto load_q = [&](int t) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int qi = (t + r) * 16 + fi;
            const __bf16* qp = qbase + (long long)(qi < N ? qi : N - 1) * q.s_n;
            qf[r][0] = *reinterpret_cast<const bf16x8*>(qp + 8 * fg);
            qf[r][1] = *reinterpret_cast<const bf16x8*>(qp + 32 + 8 * fg);
        }
    };
    auto prepare = [&](int item) {
        bh = item; b = bh / H; h = bh % H;
        qbase = q.p + b * q.s_b + h * q.s_h;
        t0 = wave * R;
        if (t0 < nqt) load_q(t0);          // in flight while K / V are staged
        fill_tiles2(Ks, k.p + b * k.s_b + h * k.s_h, k.s_n, Vs, v.p + b * v.s_b + h * v.s_h, v.s_n, N, rows_pad, tid);
    };
    auto compute = [&]() {
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    bf16x8 ones = {(__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f};
    asm volatile("" : "+v"(ones));          // four live registers: hipcc otherwise keeps one and rebuilds the operand (3 moves) in front of every row-sum MFMA
    const float c = scale_log2e;            // > 0 (checked on the host): max() commutes with the scaling
    const float inv_c = 1.0f / c;
    while (t0 < nqt) {
        // Softmax bookkeeping is kept off the VALU, which is the busy pipe of this kernel (measured: 12 VALU
        // instructions per MFMA before, VALU 55-60 % busy vs MFMA 16 %):
        //  * the row sums come from one extra MFMA per step against an all-ones operand (l = 1^T P^T), not from adds;
        //  * the reference maximum is LAZY: it is only raised (and the accumulators rescaled) when some score
        //    exceeds it by more than 2^8 -- exp2(s - mref) <= 256 is harmless in f32 / bf16 -- so after the first
        //    step the rescale and its cross-lane reduction almost never run (wave-uniform branch on a ballot);
        //  * scale and -mref are folded into the exp2 argument as one FMA.
        float mref[R], thr[R];              // thr = (mref + 8) / c: the lazy test on the raw score maximum (one compare a step)
        f32x4 acc[R][4], accl[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { mref[r] = -INFINITY; thr[r] = -INFINITY; accl[r] = z4; acc[r][0] = acc[r][1] = acc[r][2] = acc[r][3] = z4; }
        for (int s = 0; s < nks; ++s) {
            bf16x8 kf[2][2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                kf[hh][0] = row_frag(Ks, s * 32 + hh * 16, 0, fi, fg);
                kf[hh][1] = row_frag(Ks, s * 32 + hh * 16, 1, fi, fg);
            }
            bf16x8 pb[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                f32x4 st[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    st[hh] = MFMA(kf[hh][0], qf[r][0], z4);
                    st[hh] = MFMA(kf[hh][1], qf[r][1], st[hh]);
                }
                if (s == nks - 1) {            // only the last 32-key step holds padding keys
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (s * 32 + hh * 16 + 4 * fg + e >= N) st[hh][e] = -INFINITY;
                }
                float mloc = fmaxf(fmaxf(st[0][0], st[0][1]), st[0][2]);
                mloc = fmaxf(fmaxf(mloc, st[0][3]), st[1][0]);
                mloc = fmaxf(fmaxf(mloc, st[1][1]), st[1][2]);
                mloc = fmaxf(mloc, st[1][3]);
                if (__builtin_amdgcn_ballot_w64(mloc > thr[r]) != 0) {
                    const float m_new = fmaxf(mref[r], groups_max(mloc) * c);
                    const float alpha = __builtin_amdgcn_exp2f(mref[r] - m_new);
#pragma unroll
                    for (int fd = 0; fd < 4; ++fd) acc[r][fd] *= alpha;
                    accl[r] *= alpha;
                    mref[r] = m_new;
                    thr[r] = (m_new + 8.0f) * inv_c;
                }
                const float nm = -mref[r];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                    for (int e = 0; e < 4; ++e) st[hh][e] = __builtin_amdgcn_exp2f(fmaf(st[hh][e], c, nm));
                pb[r] = pack8(st[0], st[1]);
                accl[r] = MFMA(ones, pb[r], accl[r]);          // softmax denominators: of the UNDROPPED probabilities
                if (DROP && drop_t) {                          // nn.Dropout on the attention matrix (vit.py:60): zero P entries for P.V
                    const unsigned hrow = drop_row((unsigned)(bh * N + (t0 + r) * 16 + fi), drop_seed);
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (!drop_keep(hrow, (unsigned)(s * 32 + hh * 16 + 4 * fg + e), drop_t)) st[hh][e] = 0.f;
                    pb[r] = pack8(st[0], st[1]);
                }
            }
#pragma unroll
            for (int fd = 0; fd < 4; ++fd) {
                const bf16x8 vf = tr_frag(Vs, s * 32, fd * 16, fi, fg);
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r][fd] = MFMA(vf, pb[r], acc[r][fd]);
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int qi = (t0 + r) * 16 + fi;
            const float ls = accl[r][0];        // every row of 1^T P^T is the same sum
            const float inv = inv_keep / ls;    // kept entries are scaled by 1 / (1 - p)
            // [round 6] a lane holds 4 columns of each of the row's four 16-column blocks: lane pairs (fg, fg ^ 1) exchange halves
            // (v_permlane16_swap: odd 16-lane rows of the first operand <-> even rows of the second) so that every lane stores TWO 16-byte
            // pieces -- 8 consecutive columns of two blocks -- instead of four 8-byte ones (MI355X guide T21: the store tail is issue-bound)
            const bf16x8 x01 = pack8(acc[r][0] * inv, acc[r][1] * inv), x23 = pack8(acc[r][2] * inv, acc[r][3] * inv);
            const q4u xa = __builtin_bit_cast(q4u, x01), xb = __builtin_bit_cast(q4u, x23);
            unsigned pa[4], pb2[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                auto sw = __builtin_amdgcn_permlane16_swap(xa[d], xb[d], false, false);
                pa[d] = (unsigned)sw[0]; pb2[d] = (unsigned)sw[1];
            }
            if (qi < N) {
                __bf16* op = o.p + b * o.s_b + h * o.s_h + (long long)qi * o.s_n + ((fg & 1) ? 32 : 0) + 4 * (fg & 2);
                *reinterpret_cast<q4u*>(op) = q4u{pa[0], pa[1], pb2[0], pb2[1]};
                *reinterpret_cast<q4u*>(op + 16) = q4u{pa[2], pa[3], pb2[2], pb2[3]};
                if (fg == 0) lse[(long long)bh * N + qi] = (mref[r] + log2f(ls)) * LN2;
            }
        }
        t0 += AT_WAVES * R;
        if (t0 < nqt) load_q(t0);
    }
    };
    prepare(blockIdx.x);
    __syncthreads();
    compute();
}

template <int R, bool DROP = true>      // DROP: see attn_fwd_kernel
__global__ __launch_bounds__(AT_THREADS) void attn_bwd_dq_kernel(BHND q, BHND k, BHND v, BHND o, BHND dout,
                                                           const float* __restrict__ lse, float* __restrict__ delta,
                                                           BHND dq, int H, int N, float scale, unsigned drop_t,
                                                           unsigned drop_seed, float inv_keep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, fg = lane >> 4;
    const int rows_pad = ((N + 31) >> 5) << 5;
    char* Ks = smem;
    char* Vs = Ks + rows_pad * AT_LD;
    const int nqt = (N + 15) >> 4, nks = rows_pad >> 5;
    const float scale_log2e = scale * LOG2E;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    int bh = 0, b = 0, h = 0, t0 = 0;

    bf16x8 qf[R][2], df[R][2];
    float dl[R], l2[R];
    auto load_rows = [&](int t) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int qi = (t + r) * 16 + fi;
            const int qrow = qi < N ? qi : N - 1;
            const __bf16* qp = q.p + b * q.s_b + h * q.s_h + (long long)qrow * q.s_n;
            const __bf16* dop = dout.p + b * dout.s_b + h * dout.s_h + (long long)qrow * dout.s_n;
            const __bf16* op = o.p + b * o.s_b + h * o.s_h + (long long)qrow * o.s_n;
            qf[r][0] = *reinterpret_cast<const bf16x8*>(qp + 8 * fg);
            qf[r][1] = *reinterpret_cast<const bf16x8*>(qp + 32 + 8 * fg);
            df[r][0] = *reinterpret_cast<const bf16x8*>(dop + 8 * fg);
            df[r][1] = *reinterpret_cast<const bf16x8*>(dop + 32 + 8 * fg);
            const bf16x8 of0 = *reinterpret_cast<const bf16x8*>(op + 8 * fg);
            const bf16x8 of1 = *reinterpret_cast<const bf16x8*>(op + 32 + 8 * fg);
            l2[r] = -lse[(long long)bh * N + qrow] * LOG2E;      // negated: the exp2 argument is one FMA
            dl[r] = groups_sum(dot8(df[r][0], of0) + dot8(df[r][1], of1));
            if (qi < N && fg == 0) delta[(long long)bh * N + qi] = dl[r];
        }
    };
    auto prepare = [&](int item) {
        bh = item; b = bh / H; h = bh % H;
        t0 = wave * R;
        if (t0 < nqt) load_rows(t0);
        fill_tiles2(Ks, k.p + b * k.s_b + h * k.s_h, k.s_n, Vs, v.p + b * v.s_b + h * v.s_h, v.s_n, N, rows_pad, tid);
    };
    auto compute = [&]() {
    while (t0 < nqt) {
        f32x4 acc[R][4];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r][0] = acc[r][1] = acc[r][2] = acc[r][3] = z4;
        for (int s = 0; s < nks; ++s) {
            bf16x8 dsb[R];
            f32x4 ds[R][2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int row0 = s * 32 + hh * 16;
                const bf16x8 k0 = row_frag(Ks, row0, 0, fi, fg), k1 = row_frag(Ks, row0, 1, fi, fg);
                const bf16x8 v0 = row_frag(Vs, row0, 0, fi, fg), v1 = row_frag(Vs, row0, 1, fi, fg);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    f32x4 st = MFMA(k0, qf[r][0], z4);
                    st = MFMA(k1, qf[r][1], st);
                    f32x4 dp = MFMA(v0, df[r][0], z4);
                    dp = MFMA(v1, df[r][1], dp);
                    if (DROP && drop_t) {             // dP = dP_dropped * keep / (1 - p)
                        const unsigned hrow = drop_row((unsigned)(bh * N + (t0 + r) * 16 + fi), drop_seed);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            dp[e] = drop_keep(hrow, (unsigned)(row0 + 4 * fg + e), drop_t) ? dp[e] * inv_keep : 0.f;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e)       // the common factor `scale` of dS is applied once, to dQ
                        ds[r][hh][e] = __builtin_amdgcn_exp2f(fmaf(st[e], scale_log2e, l2[r])) * (dp[e] - dl[r]);
                    if (s == nks - 1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (row0 + 4 * fg + e >= N) ds[r][hh][e] = 0.f;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) dsb[r] = pack8(ds[r][0], ds[r][1]);
#pragma unroll
            for (int fd = 0; fd < 4; ++fd) {
                const bf16x8 kt = tr_frag(Ks, s * 32, fd * 16, fi, fg);
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r][fd] = MFMA(kt, dsb[r], acc[r][fd]);
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int qi = (t0 + r) * 16 + fi;
            store_rows16<4>(dq.p + b * dq.s_b + h * dq.s_h + (long long)(qi < N ? qi : 0) * dq.s_n, acc[r], scale, fg, qi < N);
        }
        t0 += AT_WAVES * R;
        if (t0 < nqt) load_rows(t0);
    }
    };
    prepare(blockIdx.x);
    __syncthreads();
    compute();
}

template <int R, bool DROP = true>      // DROP: see attn_fwd_kernel
__global__ __launch_bounds__(AT_THREADS) void attn_bwd_dkv_kernel(BHND q, BHND k, BHND v, BHND dout,
                                                            const float* __restrict__ lse, const float* __restrict__ delta,
                                                            BHND dk, BHND dv, int H, int N, float scale, unsigned drop_t,
                                                            unsigned drop_seed, float inv_keep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, fg = lane >> 4;
    const int rows_pad = ((N + 31) >> 5) << 5;
    char* Qs = smem;
    char* Ds = Qs + rows_pad * AT_LD;
    float* lse_s = reinterpret_cast<float*>(Qs + 2 * rows_pad * AT_LD);
    float* del_s = lse_s + rows_pad;
    unsigned* hq_s = reinterpret_cast<unsigned*>(del_s + rows_pad);     // dropout row hashes of the query rows
    const int nkt = (N + 15) >> 4, nqs = rows_pad >> 5;
    const float scale_log2e = scale * LOG2E;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    int bh = 0, b = 0, h = 0, t0 = 0;

    bf16x8 kf[R][2], vf[R][2];
    auto load_rows = [&](int t) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int ki = (t + r) * 16 + fi;
            const int krow = ki < N ? ki : N - 1;
            const __bf16* kp = k.p + b * k.s_b + h * k.s_h + (long long)krow * k.s_n;
            const __bf16* vp = v.p + b * v.s_b + h * v.s_h + (long long)krow * v.s_n;
            kf[r][0] = *reinterpret_cast<const bf16x8*>(kp + 8 * fg);
            kf[r][1] = *reinterpret_cast<const bf16x8*>(kp + 32 + 8 * fg);
            vf[r][0] = *reinterpret_cast<const bf16x8*>(vp + 8 * fg);
            vf[r][1] = *reinterpret_cast<const bf16x8*>(vp + 32 + 8 * fg);
        }
    };
    auto prepare = [&](int item) {
        bh = item; b = bh / H; h = bh % H;
        t0 = wave * R;
        if (t0 < nkt) load_rows(t0);
        float lv = 0.f, dv_ = 0.f;     // rows_pad <= 480 < AT_THREADS: one row of lse / delta per thread
        if (tid < N) { lv = -lse[(long long)bh * N + tid] * LOG2E; dv_ = -delta[(long long)bh * N + tid]; }   // negated (FMA / add forms)
        fill_tiles2(Qs, q.p + b * q.s_b + h * q.s_h, q.s_n, Ds, dout.p + b * dout.s_b + h * dout.s_h, dout.s_n, N, rows_pad, tid);
        if (tid < rows_pad) { lse_s[tid] = lv; del_s[tid] = dv_; hq_s[tid] = drop_row((unsigned)(bh * N + tid), drop_seed); }
    };
    auto compute = [&]() {
    while (t0 < nkt) {
        f32x4 accK[R][4], accV[R][4];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int fd = 0; fd < 4; ++fd) { accK[r][fd] = z4; accV[r][fd] = z4; }
        for (int s = 0; s < nqs; ++s) {
            f32x4 p[R][2], ds[R][2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int row0 = s * 32 + hh * 16;
                const bf16x8 q0 = row_frag(Qs, row0, 0, fi, fg), q1 = row_frag(Qs, row0, 1, fi, fg);
                const bf16x8 d0 = row_frag(Ds, row0, 0, fi, fg), d1 = row_frag(Ds, row0, 1, fi, fg);
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + row0 + 4 * fg);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(del_s + row0 + 4 * fg);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    f32x4 st = MFMA(q0, kf[r][0], z4);     // S[q = row0+4g+e][key]
                    st = MFMA(q1, kf[r][1], st);
                    f32x4 dp = MFMA(d0, vf[r][0], z4);     // dP[q][key]
                    dp = MFMA(d1, vf[r][1], dp);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        p[r][hh][e] = __builtin_amdgcn_exp2f(fmaf(st[e], scale_log2e, l4[e]));
                        if (DROP && drop_t) {       // dV takes P * keep / (1 - p); dP = dP_dropped * keep / (1 - p)
                            const float km = drop_keep(hq_s[row0 + 4 * fg + e], (unsigned)((t0 + r) * 16 + fi), drop_t) ? inv_keep : 0.f;
                            ds[r][hh][e] = p[r][hh][e] * (dp[e] * km + d4[e]);
                            p[r][hh][e] *= km;
                        } else {
                            ds[r][hh][e] = p[r][hh][e] * (dp[e] + d4[e]);  // `scale` is applied once, to dK
                        }
                    }
                    if (s == nqs - 1) {                    // padding query rows only exist in the last step
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (row0 + 4 * fg + e >= N) { p[r][hh][e] = 0.f; ds[r][hh][e] = 0.f; }
                    }
                }
            }
            bf16x8 pb[R], dsb[R];
#pragma unroll
            for (int r = 0; r < R; ++r) { pb[r] = pack8(p[r][0], p[r][1]); dsb[r] = pack8(ds[r][0], ds[r][1]); }
#pragma unroll
            for (int fd = 0; fd < 4; ++fd) {
                const bf16x8 dt = tr_frag(Ds, s * 32, fd * 16, fi, fg);
                const bf16x8 qt = tr_frag(Qs, s * 32, fd * 16, fi, fg);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    accV[r][fd] = MFMA(dt, pb[r], accV[r][fd]);    // dV^T[d][key]
                    accK[r][fd] = MFMA(qt, dsb[r], accK[r][fd]);   // dK^T[d][key]
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int ki = (t0 + r) * 16 + fi;
            store_rows16<4>(dk.p + b * dk.s_b + h * dk.s_h + (long long)(ki < N ? ki : 0) * dk.s_n, accK[r], scale, fg, ki < N);
            store_rows16<4>(dv.p + b * dv.s_b + h * dv.s_h + (long long)(ki < N ? ki : 0) * dv.s_n, accV[r], 1.f, fg, ki < N);
        }
        t0 += AT_WAVES * R;
        if (t0 < nkt) load_rows(t0);
    }
    };
    prepare(blockIdx.x);
    __syncthreads();
    compute();
}

// q/k normalisation of NaViT (na_vit.py:93-101): y = x / max(||x||_2, 1e-12) * sqrt(d) * gamma[h, :] per (token, head).
// x viewed (T, H, d) with token stride ld; 16 lanes per (token, head), 4 elements per lane and 64-column chunk (NC = ceil(d / 64) chunks;
// d % 4 == 0, d <= 256: the reference takes any dim_head, na_vit.py:119; lanes past d in the last chunk carry zeros).
template <typename T, int NC>
__global__ __launch_bounds__(256) void rmsnorm_heads_fwd_kernel(const T* __restrict__ x, const T* __restrict__ gamma, T* __restrict__ y,
                                                                 float* __restrict__ rnorm, long long pairs, int H, int d, float sqrt_d,
                                                                 long long ldx, long long ldy) {
    const int sub = threadIdx.x & 15;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    for (long long pr = (long long)blockIdx.x * 16 + (threadIdx.x >> 4); pr < pairs; pr += (long long)gridDim.x * 16) {
        const long long t = pr / H; const int h = (int)(pr % H);
        f32x4 v[NC];
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int e = c * 64 + sub * 4;
            v[c] = e < d ? load4<T>(x + t * ldx + (long long)h * d + e) : z4;
            ss += v[c][0] * v[c][0] + v[c][1] * v[c][1] + v[c][2] * v[c][2] + v[c][3] * v[c][3];
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 16);
        const float rn = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int e = c * 64 + sub * 4;
            if (e < d) {
                const f32x4 g = load4<T>(gamma + (long long)h * d + e);
                store4<T>(y + t * ldy + (long long)h * d + e, v[c] * g * (rn * sqrt_d));
            }
        }
        if (sub == 0) rnorm[pr] = rn;
    }
}
// dx = s*rn*(g*dy - xhat * sum(g*dy*xhat)), xhat = x*rn, s = sqrt(d); dgamma[h,:] += dy * xhat * s (partials per block, rows of 64 NC floats)
template <typename T, int NC>
__global__ __launch_bounds__(256) void rmsnorm_heads_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ gamma,
                                                                 const float* __restrict__ rnorm, T* __restrict__ dx, float* __restrict__ partials,
                                                                 long long pairs, int H, int d, float sqrt_d, long long lddy, long long ldx, long long lddx) {
    const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
    // each 16-lane group accumulates dgamma for the heads it meets; heads cycle with period H over pairs, so a
    // group that strides by (gridDim*16) pairs keeps a fixed head only if that stride is a multiple of H: enforce it
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 accg[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) accg[c] = z4;
    const long long stride = (long long)gridDim.x * 16;
    const long long first = (long long)blockIdx.x * 16 + grp;
    const int h = (int)(first % H);
    for (long long pr = first; pr < pairs; pr += stride) {
        const long long t = pr / H;
        const float rn = rnorm[pr];
        f32x4 dv[NC], xh[NC], gd[NC];
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int e = c * 64 + sub * 4;
            const bool act = e < d;
            dv[c] = act ? load4<T>(dy + t * lddy + (long long)h * d + e) : z4;
            const f32x4 v = act ? load4<T>(x + t * ldx + (long long)h * d + e) : z4;
            const f32x4 g = act ? load4<T>(gamma + (long long)h * d + e) : z4;
            xh[c] = v * rn;
            gd[c] = g * dv[c];
            dot += gd[c][0] * xh[c][0] + gd[c][1] * xh[c][1] + gd[c][2] * xh[c][2] + gd[c][3] * xh[c][3];
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 16);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int e = c * 64 + sub * 4;
            if (e < d) store4<T>(dx + t * lddx + (long long)h * d + e, (gd[c] - xh[c] * dot) * (rn * sqrt_d));
            accg[c] += dv[c] * xh[c] * sqrt_d;
        }
    }
    // partials[(blockIdx*16 + grp)][64 NC] ; row's head = (blockIdx*16+grp) % H  (stride % H == 0 is guaranteed by the host)
#pragma unroll
    for (int c = 0; c < NC; ++c)
        *reinterpret_cast<f32x4*>(partials + ((long long)blockIdx.x * 16 + grp) * (64 * NC) + c * 64 + sub * 4) = accg[c];
}
// dgamma[h][c] = sum over partial rows r with r % H == h.  64 columns x 16 row phases per (head, chunk): the list is thousands of
// rows long (2048 per head at NaViT sizes) and a single wave walking it paid one memory latency per row (0.48 ms per
// call, 17 % of the NaViT step before this).
template <typename T>
__global__ __launch_bounds__(1024) void rmsnorm_heads_dgamma_kernel(const float* __restrict__ partials, long long nrows, int H, int d, int dpad,
                                                                     T* __restrict__ dgamma) {
    __shared__ float red[16][64];
    const int h = blockIdx.x, c = blockIdx.y * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
    float s = 0.f;
    for (long long r = h + (long long)ph * H; r < nrows; r += (long long)16 * H) s += partials[r * dpad + c];
    red[ph][threadIdx.x & 63] = s;
    __syncthreads();
    if (ph == 0 && c < d) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][threadIdx.x & 63];
        dgamma[(long long)h * d + c] = from_f32<T>(t);
    }
}

// ---- materialising pieces: row softmax and its backward (one wave per row, any cols) ----
template <typename T>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const T* __restrict__ s, T* __restrict__ p, long long rows,
                                                           int cols, float scale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long long row = (long long)blockIdx.x * 4 + wave; row < rows; row += (long long)gridDim.x * 4) {
        const T* sr = s + row * cols;
        T* pr = p + row * cols;
        float mx = -INFINITY;
        for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, to_f32<T>(sr[c]) * scale);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int c = lane; c < cols; c += 64) sum += __expf(to_f32<T>(sr[c]) * scale - mx);
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        for (int c = lane; c < cols; c += 64) pr[c] = from_f32<T>(__expf(to_f32<T>(sr[c]) * scale - mx) * inv);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const T* __restrict__ p, const T* __restrict__ dp,
                                                           T* __restrict__ ds, long long rows, int cols, float scale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long long row = (long long)blockIdx.x * 4 + wave; row < rows; row += (long long)gridDim.x * 4) {
        const T* pr = p + row * cols;
        const T* dr = dp + row * cols;
        T* o = ds + row * cols;
        float dot = 0.f;
        for (int c = lane; c < cols; c += 64) dot += to_f32<T>(pr[c]) * to_f32<T>(dr[c]);
        dot = wave_sum(dot);
        for (int c = lane; c < cols; c += 64) o[c] = from_f32<T>(scale * to_f32<T>(pr[c]) * (to_f32<T>(dr[c]) - dot));
    }
}

BHND to_bhnd(vitk_bhnd t) { return BHND{(__bf16*)t.p, (long long)t.s_b, (long long)t.s_h, (long long)t.s_n}; }
bool bhnd_ok(vitk_bhnd t) { return t.p && aligned16(t.p) && (t.s_b % 8 == 0) && (t.s_h % 8 == 0) && (t.s_n % 8 == 0); }

// Dynamic LDS above the 64 KiB default needs an opt-in; done once per kernel (thread-safe static
// initialisation) with the largest size the shape check admits (N <= 480).
constexpr size_t AT_MAX_LDS = (size_t)2 * 480 * AT_LD + (size_t)3 * 480 * sizeof(float);
template <typename K>
int set_lds_once(K kernel) {
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)AT_MAX_LDS);
}
#define SET_LDS(kernel, name) do { static const int rc__ = set_lds_once(kernel); \
    if (rc__ != 0) { vitk_set_error("%s: hipFuncSetAttribute(max dynamic LDS) failed: %d", name, rc__); return rc__; } } while (0)
int attn_shape_check(const char* name, int64_t B, int64_t H, int64_t N, int64_t d, float scale) {
    if (!(scale > 0.f)) VITK_FAIL(VITK_E_ARG, "%s: scale must be positive (got %g)", name, (double)scale);
    if (d != 64) VITK_FAIL(VITK_E_SHAPE, "%s: fused path needs dim_head == 64 (got %lld)", name, (long long)d);
    if (B <= 0 || H <= 0 || N <= 0 || N > 480 || B * H > 0x7fffffffLL)
        VITK_FAIL(VITK_E_SHAPE, "%s: fused path needs 1 <= N <= 480 (got %lld)", name, (long long)N);
    return 0;
}

// Query / key tiles per wave (template parameter R).  Forward: two once a single pass of one tile per wave would not
// cover the sequence (N > 128): every K / V fragment read then feeds two MFMAs and the wave has two independent softmax
// chains in flight (112 VGPRs, two workgroups per CU still fit).  Backward: one -- two tiles need 158 / 218 VGPRs, which
// halves the occupancy and measured slower (ViT-B shapes: 0.259 ms vs 0.276 / 0.283 ms).  VITK_ATTN_R_* override (tests).
int tiles_per_wave(const char* env, bool prefer2, int64_t N) {
    if (const char* e = vitk_switch(env)) return atoi(e) == 2 ? 2 : 1;
    return prefer2 && (N + 15) / 16 > AT_WAVES ? 2 : 1;
}

}  // namespace

extern "C" int vitk_attn_fwd_bf16(vitk_bhnd q, vitk_bhnd k, vitk_bhnd v, vitk_bhnd o, float* lse, int64_t B, int64_t H,
                                  int64_t N, int64_t d, float scale, void* stream) {
    return vitk_attn_fwd_bf16_drop(q, k, v, o, lse, B, H, N, d, scale, 0.f, 0u, stream);
}

extern "C" int vitk_attn_fwd_bf16_drop(vitk_bhnd q, vitk_bhnd k, vitk_bhnd v, vitk_bhnd o, float* lse, int64_t B, int64_t H,
                                       int64_t N, int64_t d, float scale, float drop_p, uint32_t drop_seed, void* stream) {
    if (int rc = attn_shape_check("attn_fwd_bf16", B, H, N, d, scale)) return rc;
    if (!(drop_p >= 0.f && drop_p < 1.f)) VITK_FAIL(VITK_E_ARG, "attn_fwd_bf16: dropout p must be in [0, 1) (got %g)", (double)drop_p);
    if (!bhnd_ok(q) || !bhnd_ok(k) || !bhnd_ok(v) || !bhnd_ok(o) || !lse)
        VITK_FAIL(VITK_E_ALIGN, "attn_fwd_bf16: tensors must be non-null, 16-byte aligned with strides %% 8 == 0");
    if (attn_pipe_supported(N, d) && (attn_pipe_mask() & 1)) {        // persistent, LDS-DMA-pipelined kernel (attention_pipe.hip)
        AttnPipeFwd a{};
        a.ns = 1; a.q[0] = q; a.k[0] = k; a.v[0] = v; a.o = o; a.lse = lse; a.B = B; a.H = H; a.N = N;
        a.scale = scale; a.drop_p = drop_p; a.drop_seed = drop_seed;
        return attn_pipe_fwd(a, stream);
    }
    const int rows_pad = (int)((N + 31) / 32 * 32);
    const int r = tiles_per_wave("VITK_ATTN_R_FWD", true, N);
#define ATTN_FWD_LAUNCH(RR, DD) do { SET_LDS((attn_fwd_kernel<RR, DD>), "attn_fwd_bf16"); \
        hipLaunchKernelGGL((attn_fwd_kernel<RR, DD>), dim3((unsigned)(B * H)), dim3(AT_THREADS), (size_t)2 * rows_pad * AT_LD, (hipStream_t)stream, to_bhnd(q), \
                to_bhnd(k), to_bhnd(v), to_bhnd(o), lse, (int)H, (int)N, scale * LOG2E, drop_thresh(drop_p), drop_seed, 1.0f / (1.0f - drop_p)); } while (0)
    const bool drop = drop_thresh(drop_p) != 0;
    if (r == 2) { if (drop) ATTN_FWD_LAUNCH(2, true); else ATTN_FWD_LAUNCH(2, false); }
    else        { if (drop) ATTN_FWD_LAUNCH(1, true); else ATTN_FWD_LAUNCH(1, false); }
#undef ATTN_FWD_LAUNCH
    VITK_CHECK_LAUNCH("attn_fwd_bf16");
    return 0;
}

extern "C" int vitk_attn_bwd_bf16(vitk_bhnd q, vitk_bhnd k, vitk_bhnd v, vitk_bhnd o, vitk_bhnd dout, const float* lse,
                                  float* delta, vitk_bhnd dq, vitk_bhnd dk, vitk_bhnd dv, int64_t B, int64_t H, int64_t N,
                                  int64_t d, float scale, void* stream) {
    return vitk_attn_bwd_bf16_drop(q, k, v, o, dout, lse, delta, dq, dk, dv, B, H, N, d, scale, 0.f, 0u, stream);
}

extern "C" int vitk_attn_bwd_bf16_drop(vitk_bhnd q, vitk_bhnd k, vitk_bhnd v, vitk_bhnd o, vitk_bhnd dout, const float* lse,
                                       float* delta, vitk_bhnd dq, vitk_bhnd dk, vitk_bhnd dv, int64_t B, int64_t H, int64_t N,
                                       int64_t d, float scale, float drop_p, uint32_t drop_seed, void* stream) {
    if (int rc = attn_shape_check("attn_bwd_bf16", B, H, N, d, scale)) return rc;
    if (!(drop_p >= 0.f && drop_p < 1.f)) VITK_FAIL(VITK_E_ARG, "attn_bwd_bf16: dropout p must be in [0, 1) (got %g)", (double)drop_p);
    if (!bhnd_ok(q) || !bhnd_ok(k) || !bhnd_ok(v) || !bhnd_ok(o) || !bhnd_ok(dout) || !bhnd_ok(dq) || !bhnd_ok(dk) || !bhnd_ok(dv) ||
        !lse || !delta)
        VITK_FAIL(VITK_E_ALIGN, "attn_bwd_bf16: tensors must be non-null, 16-byte aligned with strides %% 8 == 0");
    // per kernel: the pipelined flavour (attention_pipe.hip) where it is the faster one -- by default the dQ kernel only
    const int pmask = attn_pipe_mask();
    const bool fused = (pmask & 8) && attn_fused_bwd_supported(N, d, drop_p);
    const int pipe = attn_pipe_supported(N, d) ? (pmask >> 1) & 3 : 0;
    AttnPipeBwd pa{};
    if (pipe || fused) {
        pa.ns = 1; pa.q[0] = q; pa.k[0] = k; pa.v[0] = v; pa.dout[0] = dout; pa.o = o; pa.dq = dq; pa.dk = dk; pa.dv = dv;
        pa.lse = lse; pa.delta = delta; pa.B = B; pa.H = H; pa.N = N; pa.scale = scale; pa.drop_p = drop_p; pa.drop_seed = drop_seed;
    }
    if (fused) return attn_pipe_bwd_fused(pa, stream);
    const int rows_pad = (int)((N + 31) / 32 * 32);
    const size_t lds1 = (size_t)2 * rows_pad * AT_LD;
    const size_t lds2 = lds1 + (size_t)3 * rows_pad * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    const int r1 = tiles_per_wave("VITK_ATTN_R_DQ", false, N), r2 = tiles_per_wave("VITK_ATTN_R_DKV", false, N);
    const bool drop = drop_thresh(drop_p) != 0;
    if (pipe & 1) { if (int rc = attn_pipe_bwd(pa, stream, 1)) return rc; }
    else {
#define ATTN_DQ_LAUNCH(RR, DD) do { SET_LDS((attn_bwd_dq_kernel<RR, DD>), "attn_bwd_dq"); \
        hipLaunchKernelGGL((attn_bwd_dq_kernel<RR, DD>), dim3((unsigned)(B * H)), dim3(AT_THREADS), lds1, st, to_bhnd(q), to_bhnd(k), to_bhnd(v), to_bhnd(o), \
                    to_bhnd(dout), lse, delta, to_bhnd(dq), (int)H, (int)N, scale, drop_thresh(drop_p), drop_seed, 1.0f / (1.0f - drop_p)); } while (0)
        if (r1 == 2) { if (drop) ATTN_DQ_LAUNCH(2, true); else ATTN_DQ_LAUNCH(2, false); }
        else         { if (drop) ATTN_DQ_LAUNCH(1, true); else ATTN_DQ_LAUNCH(1, false); }
#undef ATTN_DQ_LAUNCH
        VITK_CHECK_LAUNCH("attn_bwd_dq");
    }
    if (pipe & 2) return attn_pipe_bwd(pa, stream, 2);
#define ATTN_DKV_LAUNCH(RR, DD) do { SET_LDS((attn_bwd_dkv_kernel<RR, DD>), "attn_bwd_dkv"); \
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<RR, DD>), dim3((unsigned)(B * H)), dim3(AT_THREADS), lds2, st, to_bhnd(q), to_bhnd(k), to_bhnd(v), to_bhnd(dout), \
                lse, delta, to_bhnd(dk), to_bhnd(dv), (int)H, (int)N, scale, drop_thresh(drop_p), drop_seed, 1.0f / (1.0f - drop_p)); } while (0)
    if (r2 == 2) { if (drop) ATTN_DKV_LAUNCH(2, true); else ATTN_DKV_LAUNCH(2, false); }
    else         { if (drop) ATTN_DKV_LAUNCH(1, true); else ATTN_DKV_LAUNCH(1, false); }
#undef ATTN_DKV_LAUNCH
    VITK_CHECK_LAUNCH("attn_bwd_dkv");
    return 0;
}

// f32-accurate flavour (validation mode): operands as hi + lo 16-bit terms (vitk_split2), f32 outputs -- attention_pipe.hip, NS = 2
extern "C" int vitk_attn_fwd_x2(vitk_bhnd q_hi, vitk_bhnd q_lo, vitk_bhnd k_hi, vitk_bhnd k_lo, vitk_bhnd v_hi, vitk_bhnd v_lo, vitk_bhnd o_f32,
                                float* lse, int64_t B, int64_t H, int64_t N, int64_t d, float scale, void* stream) {
    if (int rc = attn_shape_check("attn_fwd_x2", B, H, N, d, scale)) return rc;
    if (!attn_pipe_supported(N, d)) VITK_FAIL(VITK_E_SHAPE, "attn_fwd_x2: needs 32 < N <= 224 (got %lld)", (long long)N);
    if (!bhnd_ok(q_hi) || !bhnd_ok(q_lo) || !bhnd_ok(k_hi) || !bhnd_ok(k_lo) || !bhnd_ok(v_hi) || !bhnd_ok(v_lo) || !bhnd_ok(o_f32) || !lse)
        VITK_FAIL(VITK_E_ALIGN, "attn_fwd_x2: tensors must be non-null, 16-byte aligned with strides %% 8 == 0");
    AttnPipeFwd a{};
    a.ns = 2; a.q[0] = q_hi; a.q[1] = q_lo; a.k[0] = k_hi; a.k[1] = k_lo; a.v[0] = v_hi; a.v[1] = v_lo; a.o = o_f32; a.lse = lse;
    a.B = B; a.H = H; a.N = N; a.scale = scale; a.drop_p = 0.f; a.drop_seed = 0;
    return attn_pipe_fwd(a, stream);
}

extern "C" int vitk_attn_bwd_x2(vitk_bhnd q_hi, vitk_bhnd q_lo, vitk_bhnd k_hi, vitk_bhnd k_lo, vitk_bhnd v_hi, vitk_bhnd v_lo, vitk_bhnd o_f32,
                                vitk_bhnd do_hi, vitk_bhnd do_lo, const float* lse, float* delta, vitk_bhnd dq_f32, vitk_bhnd dk_f32,
                                vitk_bhnd dv_f32, int64_t B, int64_t H, int64_t N, int64_t d, float scale, void* stream) {
    if (int rc = attn_shape_check("attn_bwd_x2", B, H, N, d, scale)) return rc;
    if (!attn_pipe_supported(N, d)) VITK_FAIL(VITK_E_SHAPE, "attn_bwd_x2: needs 32 < N <= 224 (got %lld)", (long long)N);
    if (!bhnd_ok(q_hi) || !bhnd_ok(q_lo) || !bhnd_ok(k_hi) || !bhnd_ok(k_lo) || !bhnd_ok(v_hi) || !bhnd_ok(v_lo) || !bhnd_ok(o_f32) ||
        !bhnd_ok(do_hi) || !bhnd_ok(do_lo) || !bhnd_ok(dq_f32) || !bhnd_ok(dk_f32) || !bhnd_ok(dv_f32) || !lse || !delta)
        VITK_FAIL(VITK_E_ALIGN, "attn_bwd_x2: tensors must be non-null, 16-byte aligned with strides %% 8 == 0");
    AttnPipeBwd a{};
    a.ns = 2; a.q[0] = q_hi; a.q[1] = q_lo; a.k[0] = k_hi; a.k[1] = k_lo; a.v[0] = v_hi; a.v[1] = v_lo; a.dout[0] = do_hi; a.dout[1] = do_lo;
    a.o = o_f32; a.dq = dq_f32; a.dk = dk_f32; a.dv = dv_f32; a.lse = lse; a.delta = delta;
    a.B = B; a.H = H; a.N = N; a.scale = scale; a.drop_p = 0.f; a.drop_seed = 0;
    return attn_pipe_bwd(a, stream, 3);
}

extern "C" int vitk_softmax_fwd(const void* s, void* p, int dt, int64_t rows, int64_t cols, float scale, void* stream) {
    if (!s || !p) VITK_FAIL(VITK_E_ARG, "softmax_fwd: null pointer");
    if (rows <= 0 || cols <= 0 || cols > 0x7fffffff) VITK_FAIL(VITK_E_SHAPE, "softmax_fwd: empty");
    long long blocks = (rows + 3) / 4; if (blocks > 16384) blocks = 16384;
    VITK_DISPATCH_DT(dt, T, hipLaunchKernelGGL((softmax_fwd_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                                                (const T*)s, (T*)p, (long long)rows, (int)cols, scale));
    VITK_CHECK_LAUNCH("softmax_fwd");
    return 0;
}

extern "C" int vitk_softmax_bwd(const void* p, const void* dp, void* ds, int dt, int64_t rows, int64_t cols, float scale,
                                void* stream) {
    if (!p || !dp || !ds) VITK_FAIL(VITK_E_ARG, "softmax_bwd: null pointer");
    if (rows <= 0 || cols <= 0 || cols > 0x7fffffff) VITK_FAIL(VITK_E_SHAPE, "softmax_bwd: empty");
    long long blocks = (rows + 3) / 4; if (blocks > 16384) blocks = 16384;
    VITK_DISPATCH_DT(dt, T, hipLaunchKernelGGL((softmax_bwd_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                                                (const T*)p, (const T*)dp, (T*)ds, (long long)rows, (int)cols, scale));
    VITK_CHECK_LAUNCH("softmax_bwd");
    return 0;
}

extern "C" int64_t vitk_rmsnorm_heads_rows(int64_t T, int64_t H) {
    // partial rows of the backward kernel: blocks * 16 with (blocks * 16) % H == 0 so every 16-lane group keeps one head
    const int64_t pairs = T * H;
    int64_t blocks = (pairs + 15) / 16;
    if (blocks > 2048) blocks = 2048;
    int64_t step = H;                      // smallest b with (16 b) % H == 0 is H / gcd(16, H)
    for (int64_t g = 16; g > 0; g >>= 1) if (H % g == 0) { step = H / g; break; }
    blocks = (blocks + step - 1) / step * step;
    return blocks * 16;
}

#define VITK_RMS_NC(nc, ...) do { switch (nc) { case 1: { constexpr int NC_ = 1; __VA_ARGS__; } break; case 2: { constexpr int NC_ = 2; __VA_ARGS__; } break; \
    case 3: { constexpr int NC_ = 3; __VA_ARGS__; } break; default: { constexpr int NC_ = 4; __VA_ARGS__; } break; } } while (0)

extern "C" int vitk_rmsnorm_heads_fwd(const void* x, int64_t ldx, const void* gamma, void* y, int64_t ldy, float* rnorm, int dt,
                                      int64_t T, int64_t H, int64_t d, void* stream) {
    if (!x || !gamma || !y || !rnorm) VITK_FAIL(VITK_E_ARG, "rmsnorm_heads_fwd: null pointer");
    if (d <= 0 || d > 256 || (d & 3) || T <= 0 || H <= 0 || (ldx & 3) || (ldy & 3)) VITK_FAIL(VITK_E_SHAPE, "rmsnorm_heads_fwd: needs dim_head %% 4 == 0, dim_head <= 256, row strides %% 4 == 0");
    long long blocks = (T * H + 15) / 16; if (blocks > 4096) blocks = 4096;
    const float sd = sqrtf((float)d);
    VITK_DISPATCH_DT(dt, Tt, VITK_RMS_NC((int)((d + 63) / 64), hipLaunchKernelGGL((rmsnorm_heads_fwd_kernel<Tt, NC_>), dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, (const Tt*)x, (const Tt*)gamma, (Tt*)y, rnorm, (long long)(T * H), (int)H, (int)d, sd, (long long)ldx, (long long)ldy)));
    VITK_CHECK_LAUNCH("rmsnorm_heads_fwd");
    return 0;
}

extern "C" int vitk_rmsnorm_heads_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* gamma, const float* rnorm,
                                      void* dx, int64_t lddx, void* dgamma, float* partials, int dt, int64_t T, int64_t H, int64_t d,
                                      void* stream) {
    // partials: vitk_rmsnorm_heads_rows(T, H) rows of 64 * ceil(d / 64) floats
    if (!dy || !x || !gamma || !rnorm || !dx || !dgamma || !partials) VITK_FAIL(VITK_E_ARG, "rmsnorm_heads_bwd: null pointer");
    if (d <= 0 || d > 256 || (d & 3) || T <= 0 || H <= 0 || (lddy & 3) || (ldx & 3) || (lddx & 3)) VITK_FAIL(VITK_E_SHAPE, "rmsnorm_heads_bwd: needs dim_head %% 4 == 0, dim_head <= 256, row strides %% 4 == 0");
    const long long nrows = vitk_rmsnorm_heads_rows(T, H);
    const int nc = (int)((d + 63) / 64), dpad = 64 * nc;
    const float sd = sqrtf((float)d);
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(partials, 0, (size_t)nrows * dpad * sizeof(float), st) != hipSuccess) VITK_FAIL(1, "rmsnorm_heads_bwd: memset failed");
    VITK_DISPATCH_DT(dt, Tt, {
        VITK_RMS_NC(nc, hipLaunchKernelGGL((rmsnorm_heads_bwd_kernel<Tt, NC_>), dim3((unsigned)(nrows / 16)), dim3(256), 0, st, (const Tt*)dy, (const Tt*)x,
                           (const Tt*)gamma, rnorm, (Tt*)dx, partials, (long long)(T * H), (int)H, (int)d, sd, (long long)lddy, (long long)ldx, (long long)lddx));
        hipLaunchKernelGGL((rmsnorm_heads_dgamma_kernel<Tt>), dim3((unsigned)H, (unsigned)nc), dim3(1024), 0, st, partials, nrows, (int)H, (int)d, dpad, (Tt*)dgamma);
    });
    VITK_CHECK_LAUNCH("rmsnorm_heads_bwd");
    return 0;
}
