// attention.hip -- fused scaled-dot-product attention forward/backward for gfx950, bf16, d_head = 64.
//
// Replaces vit.py:55-63 (simple_vit.py:54-61): the three head-split rearranges, matmul(q,k^T)*scale,
// nn.Softmax, matmul(attn, v) and the head-merge rearrange -- and their autograd.  The N x N score
// matrix is never materialised; q/k/v are read in place from the merged (B, N, 3*h*d) to_qkv output
// and O / dQ,dK,dV are written in merged layouts, so no contiguity copies exist.
//
// ViT sequences are short (N = 197 for ViT-B/L at 224^2), so ONE workgroup owns one (batch, head):
// the whole K and V of that head (2 x N x 128 B) are staged once into LDS and each of the 4 waves
// walks 16-row query tiles against them.
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
// Softmax is online (running max / sum per query row, exp2 domain); backward recomputes P from
// the saved row log-sum-exp.  LDS rows are padded to 160 B: conflict-free for both the b128 row
// reads and the b64 transpose reads.
#include "common.h"

namespace {

constexpr int AT_LD = 160;  // bytes per LDS row: 64 bf16 + 32 B pad
constexpr int AT_THREADS = 512;  // 8 waves per (batch, head): query / key tiles are dealt round-robin to the waves
constexpr int AT_WAVES = AT_THREADS / 64;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

struct BHND { __bf16* p; long long s_b, s_h, s_n; };

__device__ __forceinline__ void fill_tile(char* tile, const __bf16* src, long long s_n, int N, int rows_pad, int tid) {
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int c = tid; c < rows_pad * 8; c += AT_THREADS) {
        const int row = c >> 3, col8 = c & 7;
        const bf16x8 v = row < N ? *reinterpret_cast<const bf16x8*>(src + (long long)row * s_n + col8 * 8) : zero8;
        *reinterpret_cast<bf16x8*>(tile + row * AT_LD + col8 * 16) = v;
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
__device__ __forceinline__ bf16x8 pack8(f32x4 a, f32x4 b) {
    bf16x8 r = {(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3], (__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
    return r;
}
__device__ __forceinline__ float dot8(bf16x8 a, bf16x8 b) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += (float)a[e] * (float)b[e];
    return s;
}
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

__global__ __launch_bounds__(AT_THREADS) void attn_fwd_kernel(BHND q, BHND k, BHND v, BHND o, float* __restrict__ lse,
                                                        int H, int N, float scale_log2e) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, fg = lane >> 4;
    const int rows_pad = ((N + 31) >> 5) << 5;
    char* Ks = smem;
    char* Vs = smem + rows_pad * AT_LD;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    fill_tile(Ks, k.p + b * k.s_b + h * k.s_h, k.s_n, N, rows_pad, tid);
    fill_tile(Vs, v.p + b * v.s_b + h * v.s_h, v.s_n, N, rows_pad, tid);
    __syncthreads();

    const int nqt = (N + 15) >> 4, nks = rows_pad >> 5;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    for (int qt = wave; qt < nqt; qt += AT_WAVES) {
        const int qi = qt * 16 + fi;
        const int qrow = qi < N ? qi : N - 1;
        const __bf16* qp = q.p + b * q.s_b + h * q.s_h + (long long)qrow * q.s_n;
        const bf16x8 qf0 = *reinterpret_cast<const bf16x8*>(qp + 8 * fg);
        const bf16x8 qf1 = *reinterpret_cast<const bf16x8*>(qp + 32 + 8 * fg);
        float m = -INFINITY, lsum = 0.f;
        f32x4 acc[4] = {z4, z4, z4, z4};
        for (int s = 0; s < nks; ++s) {
            f32x4 st[2];
            float mx = -INFINITY;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int row0 = s * 32 + hh * 16;
                st[hh] = MFMA(row_frag(Ks, row0, 0, fi, fg), qf0, z4);
                st[hh] = MFMA(row_frag(Ks, row0, 1, fi, fg), qf1, st[hh]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = row0 + 4 * fg + r;
                    st[hh][r] = key < N ? st[hh][r] * scale_log2e : -INFINITY;
                    mx = fmaxf(mx, st[hh][r]);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m, mx);
            const float alpha = __builtin_amdgcn_exp2f(m - m_new);
            float ps = 0.f;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int r = 0; r < 4; ++r) { st[hh][r] = __builtin_amdgcn_exp2f(st[hh][r] - m_new); ps += st[hh][r]; }
            lsum = lsum * alpha + ps;
            m = m_new;
            const bf16x8 pb = pack8(st[0], st[1]);
#pragma unroll
            for (int fd = 0; fd < 4; ++fd) {
                acc[fd] *= alpha;
                acc[fd] = MFMA(tr_frag(Vs, s * 32, fd * 16, fi, fg), pb, acc[fd]);
            }
        }
        lsum += __shfl_xor(lsum, 16, 64);
        lsum += __shfl_xor(lsum, 32, 64);
        const float inv = 1.0f / lsum;
        if (qi < N) {
            __bf16* op = o.p + b * o.s_b + h * o.s_h + (long long)qi * o.s_n + 4 * fg;
#pragma unroll
            for (int fd = 0; fd < 4; ++fd) store4<__bf16>(op + fd * 16, acc[fd] * inv);
            if (fg == 0) lse[(long long)bh * N + qi] = (m + log2f(lsum)) * LN2;
        }
    }
}

__global__ __launch_bounds__(AT_THREADS) void attn_bwd_dq_kernel(BHND q, BHND k, BHND v, BHND o, BHND dout,
                                                           const float* __restrict__ lse, float* __restrict__ delta,
                                                           BHND dq, int H, int N, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, fg = lane >> 4;
    const int rows_pad = ((N + 31) >> 5) << 5;
    char* Ks = smem;
    char* Vs = smem + rows_pad * AT_LD;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    fill_tile(Ks, k.p + b * k.s_b + h * k.s_h, k.s_n, N, rows_pad, tid);
    fill_tile(Vs, v.p + b * v.s_b + h * v.s_h, v.s_n, N, rows_pad, tid);
    __syncthreads();

    const int nqt = (N + 15) >> 4, nks = rows_pad >> 5;
    const float scale_log2e = scale * LOG2E;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    for (int qt = wave; qt < nqt; qt += AT_WAVES) {
        const int qi = qt * 16 + fi;
        const int qrow = qi < N ? qi : N - 1;
        const __bf16* qp = q.p + b * q.s_b + h * q.s_h + (long long)qrow * q.s_n;
        const __bf16* dop = dout.p + b * dout.s_b + h * dout.s_h + (long long)qrow * dout.s_n;
        const __bf16* op = o.p + b * o.s_b + h * o.s_h + (long long)qrow * o.s_n;
        const bf16x8 qf0 = *reinterpret_cast<const bf16x8*>(qp + 8 * fg);
        const bf16x8 qf1 = *reinterpret_cast<const bf16x8*>(qp + 32 + 8 * fg);
        const bf16x8 df0 = *reinterpret_cast<const bf16x8*>(dop + 8 * fg);
        const bf16x8 df1 = *reinterpret_cast<const bf16x8*>(dop + 32 + 8 * fg);
        const bf16x8 of0 = *reinterpret_cast<const bf16x8*>(op + 8 * fg);
        const bf16x8 of1 = *reinterpret_cast<const bf16x8*>(op + 32 + 8 * fg);
        float dl = dot8(df0, of0) + dot8(df1, of1);
        dl += __shfl_xor(dl, 16, 64);
        dl += __shfl_xor(dl, 32, 64);
        if (qi < N && fg == 0) delta[(long long)bh * N + qi] = dl;
        const float l2 = lse[(long long)bh * N + qrow] * LOG2E;
        f32x4 acc[4] = {z4, z4, z4, z4};
        for (int s = 0; s < nks; ++s) {
            f32x4 ds[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int row0 = s * 32 + hh * 16;
                f32x4 st = MFMA(row_frag(Ks, row0, 0, fi, fg), qf0, z4);
                st = MFMA(row_frag(Ks, row0, 1, fi, fg), qf1, st);
                f32x4 dp = MFMA(row_frag(Vs, row0, 0, fi, fg), df0, z4);
                dp = MFMA(row_frag(Vs, row0, 1, fi, fg), df1, dp);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = row0 + 4 * fg + r;
                    const float p = key < N ? __builtin_amdgcn_exp2f(st[r] * scale_log2e - l2) : 0.f;
                    ds[hh][r] = p * (dp[r] - dl) * scale;
                }
            }
            const bf16x8 dsb = pack8(ds[0], ds[1]);
#pragma unroll
            for (int fd = 0; fd < 4; ++fd) acc[fd] = MFMA(tr_frag(Ks, s * 32, fd * 16, fi, fg), dsb, acc[fd]);
        }
        if (qi < N) {
            __bf16* dqp = dq.p + b * dq.s_b + h * dq.s_h + (long long)qi * dq.s_n + 4 * fg;
#pragma unroll
            for (int fd = 0; fd < 4; ++fd) store4<__bf16>(dqp + fd * 16, acc[fd]);
        }
    }
}

__global__ __launch_bounds__(AT_THREADS) void attn_bwd_dkv_kernel(BHND q, BHND k, BHND v, BHND dout,
                                                            const float* __restrict__ lse, const float* __restrict__ delta,
                                                            BHND dk, BHND dv, int H, int N, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, fg = lane >> 4;
    const int rows_pad = ((N + 31) >> 5) << 5;
    char* Qs = smem;
    char* Ds = smem + rows_pad * AT_LD;
    float* lse_s = reinterpret_cast<float*>(smem + 2 * rows_pad * AT_LD);
    float* del_s = lse_s + rows_pad;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    fill_tile(Qs, q.p + b * q.s_b + h * q.s_h, q.s_n, N, rows_pad, tid);
    fill_tile(Ds, dout.p + b * dout.s_b + h * dout.s_h, dout.s_n, N, rows_pad, tid);
    for (int r = tid; r < rows_pad; r += AT_THREADS) {
        lse_s[r] = r < N ? lse[(long long)bh * N + r] * LOG2E : 0.f;
        del_s[r] = r < N ? delta[(long long)bh * N + r] : 0.f;
    }
    __syncthreads();

    const int nkt = (N + 15) >> 4, nqs = rows_pad >> 5;
    const float scale_log2e = scale * LOG2E;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    for (int kt = wave; kt < nkt; kt += AT_WAVES) {
        const int ki = kt * 16 + fi;
        const int krow = ki < N ? ki : N - 1;
        const __bf16* kp = k.p + b * k.s_b + h * k.s_h + (long long)krow * k.s_n;
        const __bf16* vp = v.p + b * v.s_b + h * v.s_h + (long long)krow * v.s_n;
        const bf16x8 kf0 = *reinterpret_cast<const bf16x8*>(kp + 8 * fg);
        const bf16x8 kf1 = *reinterpret_cast<const bf16x8*>(kp + 32 + 8 * fg);
        const bf16x8 vf0 = *reinterpret_cast<const bf16x8*>(vp + 8 * fg);
        const bf16x8 vf1 = *reinterpret_cast<const bf16x8*>(vp + 32 + 8 * fg);
        f32x4 accK[4] = {z4, z4, z4, z4}, accV[4] = {z4, z4, z4, z4};
        for (int s = 0; s < nqs; ++s) {
            f32x4 p[2], ds[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int row0 = s * 32 + hh * 16;
                f32x4 st = MFMA(row_frag(Qs, row0, 0, fi, fg), kf0, z4);   // S[q = row0+4g+r][key = ki]
                st = MFMA(row_frag(Qs, row0, 1, fi, fg), kf1, st);
                f32x4 dp = MFMA(row_frag(Ds, row0, 0, fi, fg), vf0, z4);   // dP[q][key]
                dp = MFMA(row_frag(Ds, row0, 1, fi, fg), vf1, dp);
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + row0 + 4 * fg);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(del_s + row0 + 4 * fg);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qidx = row0 + 4 * fg + r;
                    p[hh][r] = qidx < N ? __builtin_amdgcn_exp2f(st[r] * scale_log2e - l4[r]) : 0.f;
                    ds[hh][r] = p[hh][r] * (dp[r] - d4[r]) * scale;
                }
            }
            const bf16x8 pb = pack8(p[0], p[1]);
            const bf16x8 dsb = pack8(ds[0], ds[1]);
#pragma unroll
            for (int fd = 0; fd < 4; ++fd) {
                accV[fd] = MFMA(tr_frag(Ds, s * 32, fd * 16, fi, fg), pb, accV[fd]);   // dV^T[d][key]
                accK[fd] = MFMA(tr_frag(Qs, s * 32, fd * 16, fi, fg), dsb, accK[fd]);  // dK^T[d][key]
            }
        }
        if (ki < N) {
            __bf16* dkp = dk.p + b * dk.s_b + h * dk.s_h + (long long)ki * dk.s_n + 4 * fg;
            __bf16* dvp = dv.p + b * dv.s_b + h * dv.s_h + (long long)ki * dv.s_n + 4 * fg;
#pragma unroll
            for (int fd = 0; fd < 4; ++fd) { store4<__bf16>(dkp + fd * 16, accK[fd]); store4<__bf16>(dvp + fd * 16, accV[fd]); }
        }
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
constexpr size_t AT_MAX_LDS = (size_t)2 * 480 * AT_LD + (size_t)2 * 480 * sizeof(float);
template <typename K>
int set_lds_once(K kernel) {
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)AT_MAX_LDS);
}
#define SET_LDS(kernel, name) do { static const int rc__ = set_lds_once(kernel); \
    if (rc__ != 0) { vitk_set_error("%s: hipFuncSetAttribute(max dynamic LDS) failed: %d", name, rc__); return rc__; } } while (0)
int attn_shape_check(const char* name, int64_t B, int64_t H, int64_t N, int64_t d) {
    if (d != 64) VITK_FAIL(VITK_E_SHAPE, "%s: fused path needs dim_head == 64 (got %lld)", name, (long long)d);
    if (B <= 0 || H <= 0 || N <= 0 || N > 480 || B * H > 0x7fffffffLL)
        VITK_FAIL(VITK_E_SHAPE, "%s: fused path needs 1 <= N <= 480 (got %lld)", name, (long long)N);
    return 0;
}

}  // namespace

extern "C" int vitk_attn_fwd_bf16(vitk_bhnd q, vitk_bhnd k, vitk_bhnd v, vitk_bhnd o, float* lse, int64_t B, int64_t H,
                                  int64_t N, int64_t d, float scale, void* stream) {
    if (int rc = attn_shape_check("attn_fwd_bf16", B, H, N, d)) return rc;
    if (!bhnd_ok(q) || !bhnd_ok(k) || !bhnd_ok(v) || !bhnd_ok(o) || !lse)
        VITK_FAIL(VITK_E_ALIGN, "attn_fwd_bf16: tensors must be non-null, 16-byte aligned with strides %% 8 == 0");
    const int rows_pad = (int)((N + 31) / 32 * 32);
    const size_t lds = (size_t)2 * rows_pad * AT_LD;
    SET_LDS(attn_fwd_kernel, "attn_fwd_bf16");
    hipLaunchKernelGGL(attn_fwd_kernel, dim3((unsigned)(B * H)), dim3(AT_THREADS), lds, (hipStream_t)stream, to_bhnd(q), to_bhnd(k),
                       to_bhnd(v), to_bhnd(o), lse, (int)H, (int)N, scale * LOG2E);
    VITK_CHECK_LAUNCH("attn_fwd_bf16");
    return 0;
}

extern "C" int vitk_attn_bwd_bf16(vitk_bhnd q, vitk_bhnd k, vitk_bhnd v, vitk_bhnd o, vitk_bhnd dout, const float* lse,
                                  float* delta, vitk_bhnd dq, vitk_bhnd dk, vitk_bhnd dv, int64_t B, int64_t H, int64_t N,
                                  int64_t d, float scale, void* stream) {
    if (int rc = attn_shape_check("attn_bwd_bf16", B, H, N, d)) return rc;
    if (!bhnd_ok(q) || !bhnd_ok(k) || !bhnd_ok(v) || !bhnd_ok(o) || !bhnd_ok(dout) || !bhnd_ok(dq) || !bhnd_ok(dk) || !bhnd_ok(dv) ||
        !lse || !delta)
        VITK_FAIL(VITK_E_ALIGN, "attn_bwd_bf16: tensors must be non-null, 16-byte aligned with strides %% 8 == 0");
    const int rows_pad = (int)((N + 31) / 32 * 32);
    const size_t lds1 = (size_t)2 * rows_pad * AT_LD;
    const size_t lds2 = lds1 + (size_t)2 * rows_pad * sizeof(float);
    SET_LDS(attn_bwd_dq_kernel, "attn_bwd_dq");
    SET_LDS(attn_bwd_dkv_kernel, "attn_bwd_dkv");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((unsigned)(B * H)), dim3(AT_THREADS), lds1, st, to_bhnd(q), to_bhnd(k), to_bhnd(v),
                       to_bhnd(o), to_bhnd(dout), lse, delta, to_bhnd(dq), (int)H, (int)N, scale);
    VITK_CHECK_LAUNCH("attn_bwd_dq");
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3((unsigned)(B * H)), dim3(AT_THREADS), lds2, st, to_bhnd(q), to_bhnd(k), to_bhnd(v),
                       to_bhnd(dout), lse, delta, to_bhnd(dk), to_bhnd(dv), (int)H, (int)N, scale);
    VITK_CHECK_LAUNCH("attn_bwd_dkv");
    return 0;
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
