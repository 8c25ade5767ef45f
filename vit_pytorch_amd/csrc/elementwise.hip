// elementwise.hip -- data movement and element-wise kernels (HBM-bound), plus version / error API.
#include <cmath>
#include "common.h"
#include <stdlib.h>
#include <string.h>

// ---- error string (thread-local) ---------------------------------------------------------
static thread_local char g_err[512] = "";
void vitk_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* vitk_switch(const char* name) { return getenv(name); }
extern "C" int vitk_half_type(void) {
#ifdef VITK_HALF_IS_F16
    return VITK_F16;
#else
    return VITK_BF16;
#endif
}

extern "C" int vitk_version(void) { return VITK_VERSION; }
extern "C" const char* vitk_last_error(void) { return g_err; }

namespace {

constexpr int EW_THREADS = 256;
inline unsigned ew_blocks(long long work_items, int per_block = EW_THREADS) {
    long long b = (work_items + per_block - 1) / per_block;
    if (b > 16384) b = 16384;  // grid-stride beyond that
    if (b < 1) b = 1;
    return (unsigned)b;
}

// Rearrange 'b c (h p1) (w p2) -> b (h w) (p1 p2 c)' (vit.py:100): one wave per patch row, iterating
// in DESTINATION order so the write is fully coalesced; the gather reads hit L2 (each source line
// is touched by the p2*c consecutive destination elements of a patch row).
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const T* __restrict__ img, T* __restrict__ out, long long rows,
                                                        int C, int H, int W, int p1, int p2) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hp = H / p1, wp = W / p2;
    const int P = p1 * p2 * C;
    for (long long row = (long long)blockIdx.x * 4 + wave; row < rows; row += (long long)gridDim.x * 4) {
        const long long b = row / (hp * wp);
        const int hw = (int)(row % (hp * wp));
        const int ph = hw / wp, pw = hw % wp;
        const T* src = img + b * (long long)C * H * W + (long long)(ph * p1) * W + pw * p2;
        T* dst = out + row * (long long)P;
        for (int e = lane; e < P; e += 64) {
            const int c = e % C;
            const int ij = e / C;
            const int j = ij % p2, i = ij / p2;
            dst[e] = src[(long long)c * H * W + (long long)i * W + j];
        }
    }
}

// the gradient of that Rearrange with respect to the image: every pixel belongs to exactly one patch element, so this is the
// inverse scatter (dimg[b][c][ph p1 + i][pw p2 + j] = dpatch[(b, ph, pw)][(i p2 + j) C + c]) -- the reference differentiates through
// einops.Rearrange (vit.py:100) whenever the input requires a gradient (saliency maps, adversarial inputs)
template <typename T>
__global__ __launch_bounds__(256) void unpatchify_kernel(const T* __restrict__ dpatch, T* __restrict__ dimg, long long rows,
                                                          int C, int H, int W, int p1, int p2) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hp = H / p1, wp = W / p2;
    const int P = p1 * p2 * C;
    for (long long row = (long long)blockIdx.x * 4 + wave; row < rows; row += (long long)gridDim.x * 4) {
        const long long b = row / (hp * wp);
        const int hw = (int)(row % (hp * wp));
        const int ph = hw / wp, pw = hw % wp;
        T* dst = dimg + b * (long long)C * H * W + (long long)(ph * p1) * W + pw * p2;
        const T* src = dpatch + row * (long long)P;
        for (int e = lane; e < P; e += 64) {
            const int c = e % C;
            const int ij = e / C;
            const int j = ij % p2, i = ij / p2;
            dst[(long long)c * H * W + (long long)i * W + j] = src[e];
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long long n4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        f32x4 v = load4<T>(x + 4 * i);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
        store4<T>(y + 4 * i, v);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, T* __restrict__ dx,
                                                        long long n4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4 g = load4<T>(dy + 4 * i);
        const f32x4 v = load4<T>(x + 4 * i);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = g[e] * gelu_erf_grad(v[e]);
        store4<T>(dx + 4 * i, o);
    }
}

// Element-at-a-time forms for extents that are not multiples of 4 (T2T-ViT's token-to-token layers, t2t.py:45, are 147 and 1323
// wide): rare and small, so one scalar kernel each instead of tail handling in the vector kernels.
template <typename T>
__global__ __launch_bounds__(256) void gelu_fwd_scalar_kernel(const T* __restrict__ x, T* __restrict__ y, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        y[i] = from_f32<T>(gelu_erf(to_f32<T>(x[i])));
}
template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_scalar_kernel(const T* __restrict__ dy, const T* __restrict__ x, T* __restrict__ dx, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        dx[i] = from_f32<T>(to_f32<T>(dy[i]) * gelu_erf_grad(to_f32<T>(x[i])));
}
template <typename AT, typename BT, typename OT>
__global__ __launch_bounds__(256) void add_rows_scalar_kernel(const AT* __restrict__ a, const BT* __restrict__ b, const BT* __restrict__ bias,
                                                               OT* __restrict__ out, long long n, long long cols) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float v = to_f32<AT>(a[i]) + to_f32<BT>(b[i]);
        if (bias) v += to_f32<BT>(bias[i % cols]);
        out[i] = from_f32<OT>(v);
    }
}

template <typename AT, typename BT, typename OT>
__global__ __launch_bounds__(256) void add_rows_kernel(const AT* __restrict__ a, const BT* __restrict__ b,
                                                        const BT* __restrict__ bias, OT* __restrict__ out, long long rows,
                                                        int cols4) {
    const long long n4 = rows * cols4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        f32x4 v = load4<AT>(a + 4 * i) + load4<BT>(b + 4 * i);
        if (bias) v += load4<BT>(bias + 4 * (i % cols4));
        store4<OT>(out + 4 * i, v);
    }
}

template <typename XT, typename YT>
__global__ __launch_bounds__(256) void cast_kernel(const XT* __restrict__ x, YT* __restrict__ y, long long n) {
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256)
        store4<YT>(y + 4 * i, load4<XT>(x + 4 * i));
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) y[n4 * 4 + threadIdx.x] = from_f32<YT>(to_f32<XT>(x[n4 * 4 + threadIdx.x]));
}

// Many tensors in one launch (torch.autocast around a float32 model: every parameter is converted to the 16-bit compute type once per
// forward, and every 16-bit gradient back to float32 once per backward -- ~150 tensors for ViT-B, from a (D,) LayerNorm bias to a
// 2.4 M-element Linear weight; one launch per tensor would be ~150 launch latencies at the head of every step).  The pointer table
// travels as the kernel argument (CM_MAX tensors per launch); block b serves 8,192 elements of the tensor whose block range holds b.
constexpr int CM_MAX = 64;
constexpr long long CM_PER_BLOCK = 8192;
struct CastMany { const void* src[CM_MAX]; void* dst[CM_MAX]; long long n[CM_MAX]; int blk0[CM_MAX + 1]; int count; };
template <typename XT, typename YT>
__global__ __launch_bounds__(256) void cast_many_kernel(CastMany a) {
    int lo = 0, hi = a.count;                  // uniform binary search: blk0[lo] <= blockIdx.x < blk0[lo + 1]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int)blockIdx.x >= a.blk0[mid]) lo = mid; else hi = mid; }
    const XT* __restrict__ x = (const XT*)a.src[lo];
    YT* __restrict__ y = (YT*)a.dst[lo];
    const long long n = a.n[lo], n4 = n >> 2;
    const long long b0 = (long long)((int)blockIdx.x - a.blk0[lo]) * (CM_PER_BLOCK / 4);
    const long long b1 = b0 + CM_PER_BLOCK / 4 < n4 ? b0 + CM_PER_BLOCK / 4 : n4;
    for (long long i = b0 + threadIdx.x; i < b1; i += 256) store4<YT>(y + 4 * i, load4<XT>(x + 4 * i));
    if ((int)blockIdx.x == a.blk0[lo] && threadIdx.x < (n & 3)) y[n4 * 4 + threadIdx.x] = from_f32<YT>(to_f32<XT>(x[n4 * 4 + threadIdx.x]));
}

// Adam / AdamW step over one flat range (torch.optim.Adam semantics, train_vit_decorr.py:68-70,110):
//   g' = g + wd * p (coupled) | p *= 1 - lr * wd (decoupled);  m = b1 m + (1-b1) g';  v = b2 v + (1-b2) g'^2
//   p -= (lr / c1) * m / (sqrt(v) / sqrt(c2) + eps)            c1 = 1 - b1^t, c2 = 1 - b2^t
// Moments (and the optional master copy of low-precision parameters) are f32.  One pass: 2+2 B (bf16 p, g) + 8 B read
// and 2 + 8 B written per element (+ 4 + 4 with a master copy).
template <typename PT>
__global__ __launch_bounds__(256) void adam_kernel(PT* __restrict__ p, const PT* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, float* __restrict__ master, long long n, float lr,
                                                    float b1, float b2, float eps, float wd, int decoupled, float inv_c1,
                                                    float inv_sqrt_c2, float gscale) {
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        f32x4 pv = master ? *reinterpret_cast<const f32x4*>(master + 4 * i) : load4<PT>(p + 4 * i);
        f32x4 gv = load4<PT>(g + 4 * i) * gscale;
        f32x4 mv = *reinterpret_cast<const f32x4*>(m + 4 * i), vv = *reinterpret_cast<const f32x4*>(v + 4 * i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (decoupled) pv[e] *= 1.0f - lr * wd; else gv[e] = fmaf(wd, pv[e], gv[e]);
            mv[e] = fmaf(b1, mv[e], (1.0f - b1) * gv[e]);
            vv[e] = fmaf(b2, vv[e], (1.0f - b2) * gv[e] * gv[e]);
            pv[e] -= (lr * inv_c1) * mv[e] / (sqrtf(vv[e]) * inv_sqrt_c2 + eps);
        }
        *reinterpret_cast<f32x4*>(m + 4 * i) = mv;
        *reinterpret_cast<f32x4*>(v + 4 * i) = vv;
        if (master) *reinterpret_cast<f32x4*>(master + 4 * i) = pv;
        store4<PT>(p + 4 * i, pv);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long long i = n4 * 4 + threadIdx.x;
        float pv = master ? master[i] : to_f32<PT>(p[i]);
        float gv = to_f32<PT>(g[i]) * gscale;
        if (decoupled) pv *= 1.0f - lr * wd; else gv = fmaf(wd, pv, gv);
        const float mv = fmaf(b1, m[i], (1.0f - b1) * gv), vv = fmaf(b2, v[i], (1.0f - b2) * gv * gv);
        pv -= (lr * inv_c1) * mv / (sqrtf(vv) * inv_sqrt_c2 + eps);
        m[i] = mv; v[i] = vv;
        if (master) master[i] = pv;
        p[i] = from_f32<PT>(pv);
    }
}

// ---- fp8 (OCP e4m3) quantisation for vitk_gemm_nt_fp8 -----------------------------------------------------------------
// absmax: one atomicMax on the float's bit pattern per block (non-negative floats order like unsigned ints).
template <typename T>
__global__ __launch_bounds__(256) void absmax_kernel(const T* __restrict__ x, long long n, unsigned* __restrict__ out) {
    float m = 0.f;
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4 v = load4<T>(x + 4 * i);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(to_f32<T>(x[n4 * 4 + threadIdx.x])));
    m = wave_max(m);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, __builtin_bit_cast(unsigned, fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}
// out = e4m3(clamp(x * scale, -448, 448)), round to nearest even (v_cvt_pk_fp8_f32), 4 elements -> one dword
template <typename T>
__global__ __launch_bounds__(256) void quantize_fp8_kernel(const T* __restrict__ x, unsigned* __restrict__ out, long long n4,
                                                            const float* __restrict__ scale_dev, float scale_host) {
    const float sc = scale_dev ? *scale_dev : scale_host;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        f32x4 v = load4<T>(x + 4 * i) * sc;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], -448.f, 448.f);
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w, true);
        out[i] = (unsigned)w;
    }
}
// scale[0] = 448 / max(amax, tiny);  scale[1] = 1 / scale[0]   (device-side, so the step never syncs with the host)
__global__ void fp8_scale_kernel(const unsigned* __restrict__ amax_bits, float* __restrict__ scale) {
    const float a = fmaxf(__builtin_bit_cast(float, *amax_bits), 1e-12f);
    scale[0] = 448.f / a;
    scale[1] = a / 448.f;
}

__global__ __launch_bounds__(256) void dropout_keep_kernel(uint8_t* __restrict__ keep, long long rows, long long cols, unsigned thresh,
                                                            unsigned seed) {
    const long long n = rows * cols;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const unsigned r = (unsigned)(i / cols), c = (unsigned)(i % cols);
        keep[i] = drop_keep(drop_row(r, seed), c, thresh) ? 1 : 0;
    }
}

// Delayed scaling bookkeeping: slot s owns 64 amax words (bit patterns) and 2 floats {scale, 1/scale}.  After a step:
// scale = 448 / max(amax words) for every slot that recorded something, words reset to 0.
__global__ __launch_bounds__(64) void fp8_update_scales_kernel(unsigned* __restrict__ amax, float* __restrict__ scales, int nslots) {
    const int s = blockIdx.x;
    if (s >= nslots) return;
    float m = __builtin_bit_cast(float, amax[s * 64 + threadIdx.x]);
    amax[s * 64 + threadIdx.x] = 0u;
    m = wave_max(m);
    if (threadIdx.x == 0 && m > 0.f) { scales[2 * s] = 448.f / m; scales[2 * s + 1] = m / 448.f; }
}

// vitk_fp8_update_scales_fmt: the same fold with a per-slot format maximum (e4m3 448, e5m2 57344 or less for headroom)
__global__ __launch_bounds__(64) void fp8_update_scales_fmt_kernel(unsigned* __restrict__ amax, float* __restrict__ scales, int nslots,
                                                                   const float* __restrict__ fmax) {
    const int s = blockIdx.x;
    if (s >= nslots) return;
    float m = __builtin_bit_cast(float, amax[s * 64 + threadIdx.x]);
    amax[s * 64 + threadIdx.x] = 0u;
    m = wave_max(m);
    const float fm = fmax ? fmax[s] : 448.f;
    if (threadIdx.x == 0 && m > 0.f) { scales[2 * s] = fm / m; scales[2 * s + 1] = m / fm; }
}

// One pass: out8 = fp8(clamp(x * scale2[0])) under the scale decided before this step (out8 null: record only) and this step's
// max|x| into amax64[blockIdx & 63] (null: no record).  BF8: OCP e5m2 (v_cvt_pk_bf8_f32, +-57344) instead of e4m3 (+-448).
template <typename T, bool BF8>
__global__ __launch_bounds__(256) void quantize_fp8_delayed_kernel(const T* __restrict__ x, unsigned* __restrict__ out, long long n4,
                                                                    const float* __restrict__ scale2, unsigned* __restrict__ amax64) {
    const float sc = out ? scale2[0] : 0.f;
    constexpr float FMAX = BF8 ? 57344.f : 448.f;
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4 v0 = load4<T>(x + 4 * i);
        m = fmaxf(m, absmax4(v0));
        if (out) {
            f32x4 v = v0 * sc;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], -FMAX, FMAX);
            int w = 0;
            if constexpr (BF8) {
                w = __builtin_amdgcn_cvt_pk_bf8_f32(v[0], v[1], w, false);
                w = __builtin_amdgcn_cvt_pk_bf8_f32(v[2], v[3], w, true);
            } else {
                w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], w, false);
                w = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w, true);
            }
            out[i] = (unsigned)w;
        }
    }
    if (amax64) {       // uniform branch; one atomic per block, spread over the slot's 64 words (same-address atomics serialise)
        m = wave_max(m);
        __shared__ float red[4];
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) atomicMax(amax64 + (blockIdx.x & 63), __builtin_bit_cast(unsigned, fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
    }
}

template <typename XT, typename PT>
__global__ __launch_bounds__(256) void write_cls_kernel(XT* __restrict__ x, const PT* __restrict__ cls, const PT* __restrict__ pos,
                                                         long long B, long long N, int D, int ncls) {
    const long long total = B * ncls * D;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int d = (int)(i % D);
        const long long t = i / D;
        const int c = (int)(t % ncls);
        const long long b = t / ncls;
        x[(b * N + c) * D + d] = from_f32<XT>(to_f32<PT>(cls[c * D + d]) + to_f32<PT>(pos[c * D + d]));
    }
}

// out[b, d] = mean_n x[b, n, d]; thread per (b, 4 columns)
template <typename XT, typename OT>
__global__ __launch_bounds__(256) void mean_pool_fwd_kernel(const XT* __restrict__ x, OT* __restrict__ out, long long B,
                                                             int N, int D4) {
    const long long total = B * D4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long b = i / D4;
        const int c = (int)(i % D4);
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int n = 0; n < N; ++n) s += load4<XT>(x + ((b * N + n) * D4 + c) * 4);
        s *= (1.0f / (float)N);
        store4<OT>(out + i * 4, s);
    }
}
template <typename DT, typename XT>
__global__ __launch_bounds__(256) void mean_pool_bwd_kernel(const DT* __restrict__ dout, XT* __restrict__ dx, long long B,
                                                             int N, int D4) {
    const long long total = B * N * D4;
    const float inv = 1.0f / (float)N;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % D4);
        const long long b = i / ((long long)N * D4);
        store4<XT>(dx + i * 4, load4<DT>(dout + (b * D4 + c) * 4) * inv);
    }
}

// Counter-based RNG for dropout: a 64-bit mix (splitmix64 finaliser) of (seed, offset + index).
// Stateless, so forward and backward agree through the stored u8 mask and the stream of keep
// decisions is reproducible from (seed, offset) alone.
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
template <typename T>
__global__ __launch_bounds__(256) void dropout_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ mask,
                                                           long long n, float p, unsigned long long seed,
                                                           unsigned long long offset) {
    const float scale = 1.0f / (1.0f - p);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const unsigned long long r = mix64(seed ^ mix64(offset + (unsigned long long)i));
        const float u = (float)(r >> 40) * (1.0f / 16777216.0f);  // 24 random bits -> [0,1)
        const bool keep = u >= p;
        mask[i] = keep ? 1 : 0;
        y[i] = from_f32<T>(keep ? to_f32<T>(x[i]) * scale : 0.f);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void dropout_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ mask,
                                                           T* __restrict__ dx, long long n, float p) {
    const float scale = 1.0f / (1.0f - p);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        dx[i] = from_f32<T>(mask[i] ? to_f32<T>(dy[i]) * scale : 0.f);
}

// 32x32 LDS-tiled transpose
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ in, T* __restrict__ out, int rows, int cols) {
    __shared__ T tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + 8 * k][tx] = in[(long long)r * cols + c];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;
        if (r < rows && c < cols) out[(long long)c * rows + r] = tile[tx][ty + 8 * k];
    }
}

// NaViT patch order 'c (h p1) (w p2) -> (h w) (c p1 p2)' (na_vit.py:300): element e = (c*p + i)*p + j.
template <typename T>
__global__ __launch_bounds__(256) void patchify_cpp_kernel(const T* __restrict__ img, T* __restrict__ out, int C, int H, int W, int p,
                                                            long long row0, long long ld) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hp = H / p, wp = W / p;
    const int P = C * p * p;
    for (int row = blockIdx.x * 4 + wave; row < hp * wp; row += gridDim.x * 4) {
        const int ph = row / wp, pw = row % wp;
        const T* src = img + (long long)(ph * p) * W + pw * p;
        T* dst = out + (row0 + row) * ld;
        for (int e = lane; e < P; e += 64) {
            const int j = e % p;
            const int ci = e / p;
            const int i = ci % p, c = ci / p;
            dst[e] = src[(long long)c * H * W + (long long)i * W + j];
        }
    }
}

// the adjoint of patchify_cpp_kernel: dimg[c][ph p + i][pw p + j] = dpatch[(row0 + row) ld + e] (every image element belongs to exactly one patch)
template <typename T>
__global__ __launch_bounds__(256) void unpatchify_cpp_kernel(const T* __restrict__ dpatch, T* __restrict__ dimg, int C, int H, int W, int p,
                                                              long long row0, long long ld) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hp = H / p, wp = W / p;
    const int P = C * p * p;
    for (int row = blockIdx.x * 4 + wave; row < hp * wp; row += gridDim.x * 4) {
        const int ph = row / wp, pw = row % wp;
        T* dst = dimg + (long long)(ph * p) * W + pw * p;
        const T* src = dpatch + (row0 + row) * ld;
        for (int e = lane; e < P; e += 64) {
            const int j = e % p;
            const int ci = e / p;
            const int i = ci % p, c = ci / p;
            dst[(long long)c * H * W + (long long)i * W + j] = src[e];
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gather_add2_kernel(const T* __restrict__ x, const T* __restrict__ A, const int* __restrict__ ia,
                                                           const T* __restrict__ B, const int* __restrict__ ib, T* __restrict__ out,
                                                           long long Tn, int D4) {
    const long long n4 = Tn * D4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const long long t = i / D4;
        const int c = (int)(i % D4);
        const f32x4 v = load4<T>(x + i * 4) + load4<T>(A + ((long long)ia[t] * D4 + c) * 4) + load4<T>(B + ((long long)ib[t] * D4 + c) * 4);
        store4<T>(out + i * 4, v);
    }
}

template <typename GT, typename OT>
__global__ __launch_bounds__(256) void csr_rowsum_kernel(const GT* __restrict__ g, const int* __restrict__ ptr, const int* __restrict__ rows,
                                                          OT* __restrict__ out, int D4) {
    const int seg = blockIdx.x;
    const int b = ptr[seg], e = ptr[seg + 1];
    for (int c = threadIdx.x; c < D4; c += 256) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int j = b; j < e; ++j) s += load4<GT>(g + ((long long)rows[j] * D4 + c) * 4);
        store4<OT>(out + ((long long)seg * D4 + c) * 4, s);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void copy_cols_kernel(const T* __restrict__ src, long long lds_, T* __restrict__ dst, long long ldd,
                                                         long long rows, int cols_copy, int cols_dst) {
    const long long total = rows * cols_dst;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / cols_dst;
        const int c = (int)(i % cols_dst);
        dst[r * ldd + c] = c < cols_copy ? src[r * lds_ + c] : from_f32<T>(0.f);
    }
}

// out[b, i, :] = (i < F ? front[i, :] : x[b, i - F, :]) + (pos ? pos[i, :] : 0): torch.cat((front tokens, x), dim=1) (+ pos[:N])
// for every image in ONE launch (vit.py:122-127 cls + pos).  BEHIND: the extra tokens follow x -- pack([x, r]) of
// simple_vit_with_register_tokens.py:113-115: out[b, i, :] = (i < Np ? x[b, i, :] : front[i - Np, :]) (+ pos)
template <typename T, bool BEHIND>
__global__ __launch_bounds__(256) void concat_tokens_kernel(const T* __restrict__ x, const T* __restrict__ front, const T* __restrict__ pos,
                                                             T* __restrict__ out, long long B, int Np, int F, int D4) {
    const int N = Np + F;
    const long long total = B * N * D4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % D4);
        const long long r = i / D4;
        const int t = (int)(r % N);
        const long long b = r / N;
        f32x4 v;
        if constexpr (BEHIND) v = t < Np ? load4<T>(x + ((b * Np + t) * D4 + c) * 4) : load4<T>(front + ((long long)(t - Np) * D4 + c) * 4);
        else v = t < F ? load4<T>(front + ((long long)t * D4 + c) * 4) : load4<T>(x + ((b * Np + (t - F)) * D4 + c) * 4);
        if (pos) v += load4<T>(pos + ((long long)t * D4 + c) * 4);
        store4<T>(out + i * 4, v);
    }
}

// the same for a width that is not a multiple of 4 (rows are not 8-byte aligned: one element per thread)
template <typename T, bool BEHIND>
__global__ __launch_bounds__(256) void concat_tokens_any_kernel(const T* __restrict__ x, const T* __restrict__ front, const T* __restrict__ pos,
                                                                 T* __restrict__ out, long long B, int Np, int F, int D) {
    const int N = Np + F;
    const long long total = B * N * D;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % D);
        const long long r = i / D;
        const int t = (int)(r % N);
        const long long b = r / N;
        float v;
        if constexpr (BEHIND) v = t < Np ? (float)x[(b * Np + t) * D + c] : (float)front[(long long)(t - Np) * D + c];
        else v = t < F ? (float)front[(long long)t * D + c] : (float)x[(b * Np + (t - F)) * D + c];
        if (pos) v += (float)pos[(long long)t * D + c];
        out[i] = (T)v;
    }
}

// out[b, j, :] = x[b, idx[b, j], :]  (PatchDropout: vit_with_patch_dropout.py:28-32) and its adjoint dx[b, idx[b, j], :] = g[b, j, :]
// (the indices of one image are distinct, dx is zeroed by the caller)
template <typename T, bool SCATTER>
__global__ __launch_bounds__(256) void gather_tokens_kernel(const T* __restrict__ src, const int* __restrict__ idx, T* __restrict__ dst,
                                                             long long B, int Np, int Kp, int D4) {
    const long long total = B * Kp * D4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % D4);
        const long long r = i / D4;
        const int j = (int)(r % Kp);
        const long long b = r / Kp;
        const int t = idx[b * Kp + j];
        const long long full = ((b * Np + t) * D4 + c) * 4;
        if (SCATTER) store4<T>(dst + full, load4<T>(src + i * 4));
        else store4<T>(dst + i * 4, load4<T>(src + full));
    }
}

// x (f32) = hi + mid + lo with three bf16 terms (8 + 8 + 8 significant bits: exact for every normal f32 whose low terms do not
// underflow); the six products  hi.hi + hi.mid + mid.hi + hi.lo + lo.hi + mid.mid  of two split operands reproduce the f32 product
// to ~2^-22.  This kernel writes the six-block image one operand needs: block b of operand A holds part {0,0,1,0,2,1}[b], of operand
// B part {0,1,0,2,0,1}[b]; blocks are `block_stride` elements apart (K-concatenation for the NT kernel: block_stride = K,
// ld_out = 6 K; M-concatenation for the TN kernel: block_stride = M * ld_out).
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, long long ldx, __bf16* __restrict__ out, long long ldo,
                                                      long long block_stride, long long rows, int cols4, int operand_b) {
    const long long total = rows * cols4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / cols4;
        const int c = (int)(i % cols4) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + c);
        f32x4 part[3];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const __bf16 h = (__bf16)v[e];
            const float r1 = v[e] - (float)h;
            const __bf16 m = (__bf16)r1;
            const float r2 = r1 - (float)m;
            part[0][e] = (float)h; part[1][e] = (float)m; part[2][e] = (float)(__bf16)r2;
        }
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const int which = ((operand_b ? 0x010201 : 0x001021) >> (4 * (5 - b))) & 0xf;      // one hex digit per block
            store4<__bf16>(out + b * block_stride + r * ldo + c, part[which]);
        }
    }
}

}  // namespace

extern "C" int vitk_split_bf16x3(const float* x, int64_t ldx, void* out, int64_t ld_out, int64_t block_stride, int64_t rows,
                                 int64_t cols, int operand_b, void* stream) {
    if (!x || !out) VITK_FAIL(VITK_E_ARG, "split_bf16x3: null pointer");
    if (rows <= 0 || cols <= 0 || (cols & 3) || (ldx & 3) || (ld_out & 3) || (block_stride & 3) || !aligned16(x) || !aligned8(out))
        VITK_FAIL(VITK_E_SHAPE, "split_bf16x3: extents and strides must be multiples of 4");
    hipLaunchKernelGGL(split3_kernel, dim3(ew_blocks(rows * cols / 4)), dim3(256), 0, (hipStream_t)stream, x, (long long)ldx,
                       (__bf16*)out, (long long)ld_out, (long long)block_stride, (long long)rows, (int)(cols / 4), operand_b);
    VITK_CHECK_LAUNCH("split_bf16x3");
    return 0;
}

extern "C" int vitk_concat_tokens(const void* x, const void* front, const void* pos, void* out, int dt, int64_t B, int64_t Np,
                                  int64_t F, int64_t D, void* stream) {
    const bool behind = F < 0;          // F < 0: |F| extra tokens BEHIND x
    if (behind) F = -F;
    if (!x || !out || (F > 0 && !front)) VITK_FAIL(VITK_E_ARG, "concat_tokens: null pointer");
    if (B <= 0 || Np < 0 || Np + F <= 0 || D <= 0) VITK_FAIL(VITK_E_SHAPE, "concat_tokens: need a non-empty sequence");
    if (D & 3) {        // any width (ViT(dim = 30): the reference has no such constraint), one element per thread
        if (behind) {
            VITK_DISPATCH_DT(dt, T, hipLaunchKernelGGL((concat_tokens_any_kernel<T, true>), dim3(ew_blocks(B * (Np + F) * D)), dim3(256), 0, (hipStream_t)stream,
                                                        (const T*)x, (const T*)front, (const T*)pos, (T*)out, (long long)B, (int)Np, (int)F, (int)D));
        } else {
            VITK_DISPATCH_DT(dt, T, hipLaunchKernelGGL((concat_tokens_any_kernel<T, false>), dim3(ew_blocks(B * (Np + F) * D)), dim3(256), 0, (hipStream_t)stream,
                                                        (const T*)x, (const T*)front, (const T*)pos, (T*)out, (long long)B, (int)Np, (int)F, (int)D));
        }
        VITK_CHECK_LAUNCH("concat_tokens");
        return 0;
    }
    if (behind) {
        VITK_DISPATCH_DT(dt, T, hipLaunchKernelGGL((concat_tokens_kernel<T, true>), dim3(ew_blocks(B * (Np + F) * D / 4)), dim3(256), 0, (hipStream_t)stream,
                                                    (const T*)x, (const T*)front, (const T*)pos, (T*)out, (long long)B, (int)Np, (int)F, (int)(D / 4)));
    } else {
        VITK_DISPATCH_DT(dt, T, hipLaunchKernelGGL((concat_tokens_kernel<T, false>), dim3(ew_blocks(B * (Np + F) * D / 4)), dim3(256), 0, (hipStream_t)stream,
                                                    (const T*)x, (const T*)front, (const T*)pos, (T*)out, (long long)B, (int)Np, (int)F, (int)(D / 4)));
    }
    VITK_CHECK_LAUNCH("concat_tokens");
    return 0;
}

extern "C" int vitk_gather_tokens(const void* src, const int32_t* idx, void* dst, int dt, int64_t B, int64_t Np, int64_t Kp, int64_t D,
                                  int scatter, void* stream) {
    if (!src || !idx || !dst) VITK_FAIL(VITK_E_ARG, "gather_tokens: null pointer");
    if (B <= 0 || Np <= 0 || Kp <= 0 || Kp > Np || D <= 0 || (D & 3)) VITK_FAIL(VITK_E_SHAPE, "gather_tokens: need 0 < Kp <= Np and D %% 4 == 0");
    const unsigned blocks = ew_blocks(B * Kp * D / 4);
    if (scatter) {
        VITK_DISPATCH_DT(dt, T, hipLaunchKernelGGL((gather_tokens_kernel<T, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                                                    (const T*)src, (const int*)idx, (T*)dst, (long long)B, (int)Np, (int)Kp, (int)(D / 4)));
    } else {
        VITK_DISPATCH_DT(dt, T, hipLaunchKernelGGL((gather_tokens_kernel<T, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                                                    (const T*)src, (const int*)idx, (T*)dst, (long long)B, (int)Np, (int)Kp, (int)(D / 4)));
    }
    VITK_CHECK_LAUNCH("gather_tokens");
    return 0;
}

extern "C" int vitk_patchify(const void* img, void* out, int dt, int64_t B, int64_t C, int64_t H, int64_t W, int64_t p1,
                             int64_t p2, void* stream) {
    if (!img || !out) VITK_FAIL(VITK_E_ARG, "patchify: null pointer");
    if (B <= 0 || C <= 0 || p1 <= 0 || p2 <= 0 || H % p1 || W % p2) VITK_FAIL(VITK_E_SHAPE, "patchify: image not divisible by patch");
    const long long rows = B * (H / p1) * (W / p2);
    VITK_DISPATCH_DT(dt, T, hipLaunchKernelGGL((patchify_kernel<T>), dim3(ew_blocks(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                                                (const T*)img, (T*)out, rows, (int)C, (int)H, (int)W, (int)p1, (int)p2));
    VITK_CHECK_LAUNCH("patchify");
    return 0;
}

extern "C" int vitk_unpatchify(const void* dpatch, void* dimg, int dt, int64_t B, int64_t C, int64_t H, int64_t W, int64_t p1,
                               int64_t p2, void* stream) {
    if (!dpatch || !dimg) VITK_FAIL(VITK_E_ARG, "unpatchify: null pointer");
    if (B <= 0 || C <= 0 || p1 <= 0 || p2 <= 0 || H % p1 || W % p2) VITK_FAIL(VITK_E_SHAPE, "unpatchify: image not divisible by patch");
    const long long rows = B * (H / p1) * (W / p2);
    VITK_DISPATCH_DT(dt, T, hipLaunchKernelGGL((unpatchify_kernel<T>), dim3(ew_blocks(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                                                (const T*)dpatch, (T*)dimg, rows, (int)C, (int)H, (int)W, (int)p1, (int)p2));
    VITK_CHECK_LAUNCH("unpatchify");
    return 0;
}

extern "C" int vitk_gelu_fwd(const void* x, void* y, int dt, int64_t n, void* stream) {
    if (!x || !y) VITK_FAIL(VITK_E_ARG, "gelu_fwd: null pointer");
    if (n <= 0) VITK_FAIL(VITK_E_SHAPE, "gelu_fwd: empty");
    if (n & 3) {
        VITK_DISPATCH_DT(dt, T, hipLaunchKernelGGL((gelu_fwd_scalar_kernel<T>), dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream,
                                                    (const T*)x, (T*)y, (long long)n));
        VITK_CHECK_LAUNCH("gelu_fwd");
        return 0;
    }
    VITK_DISPATCH_DT(dt, T, hipLaunchKernelGGL((gelu_fwd_kernel<T>), dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream,
                                                (const T*)x, (T*)y, (long long)(n / 4)));
    VITK_CHECK_LAUNCH("gelu_fwd");
    return 0;
}
extern "C" int vitk_gelu_bwd(const void* dy, const void* x, void* dx, int dt, int64_t n, void* stream) {
    if (!dy || !x || !dx) VITK_FAIL(VITK_E_ARG, "gelu_bwd: null pointer");
    if (n <= 0) VITK_FAIL(VITK_E_SHAPE, "gelu_bwd: empty");
    if (n & 3) {
        VITK_DISPATCH_DT(dt, T, hipLaunchKernelGGL((gelu_bwd_scalar_kernel<T>), dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream,
                                                    (const T*)dy, (const T*)x, (T*)dx, (long long)n));
        VITK_CHECK_LAUNCH("gelu_bwd");
        return 0;
    }
    VITK_DISPATCH_DT(dt, T, hipLaunchKernelGGL((gelu_bwd_kernel<T>), dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream,
                                                (const T*)dy, (const T*)x, (T*)dx, (long long)(n / 4)));
    VITK_CHECK_LAUNCH("gelu_bwd");
    return 0;
}

extern "C" int vitk_add_rows(const void* a, int adt, const void* b, int bdt, const void* bias, int biasdt, void* out, int odt,
                             int64_t rows, int64_t cols, void* stream) {
    if (!a || !b || !out) VITK_FAIL(VITK_E_ARG, "add_rows: null pointer");
    if (rows <= 0 || cols <= 0) VITK_FAIL(VITK_E_SHAPE, "add_rows: empty");
    if (bias && biasdt != bdt) VITK_FAIL(VITK_E_DTYPE, "add_rows: bias dtype must equal b dtype");
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (cols & 3) == 0;
    const unsigned blocks = ew_blocks(vec ? rows * cols / 4 : rows * cols);
#define ADD_CASE(AT, BT, OT) do { \
        if (vec) hipLaunchKernelGGL((add_rows_kernel<AT, BT, OT>), dim3(blocks), dim3(256), 0, st, (const AT*)a, \
                                    (const BT*)b, (const BT*)bias, (OT*)out, (long long)rows, (int)(cols / 4)); \
        else hipLaunchKernelGGL((add_rows_scalar_kernel<AT, BT, OT>), dim3(blocks), dim3(256), 0, st, (const AT*)a, \
                                (const BT*)b, (const BT*)bias, (OT*)out, (long long)(rows * cols), (long long)cols); \
    } while (0)
    if (adt == VITK_F32 && bdt == VITK_F32 && odt == VITK_F32) ADD_CASE(float, float, float);
    else if (adt == VITK_F32 && bdt == VITK_BF16 && odt == VITK_F32) ADD_CASE(float, __bf16, float);
    else if (adt == VITK_F32 && bdt == VITK_BF16 && odt == VITK_BF16) ADD_CASE(float, __bf16, __bf16);
    else if (adt == VITK_BF16 && bdt == VITK_BF16 && odt == VITK_BF16) ADD_CASE(__bf16, __bf16, __bf16);
    else if (adt == VITK_BF16 && bdt == VITK_BF16 && odt == VITK_F32) ADD_CASE(__bf16, __bf16, float);
    else VITK_FAIL(VITK_E_DTYPE, "add_rows: unsupported dtype combination");
#undef ADD_CASE
    VITK_CHECK_LAUNCH("add_rows");
    return 0;
}

extern "C" int vitk_cast(const void* x, int xdt, void* y, int ydt, int64_t n, void* stream) {
    if (!x || !y) VITK_FAIL(VITK_E_ARG, "cast: null pointer");
    if (n <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const unsigned blocks = ew_blocks((n + 3) / 4);
    if (xdt == VITK_F32 && ydt == VITK_F32) hipLaunchKernelGGL((cast_kernel<float, float>), dim3(blocks), dim3(256), 0, st, (const float*)x, (float*)y, (long long)n);
    else if (xdt == VITK_F32 && ydt == VITK_BF16) hipLaunchKernelGGL((cast_kernel<float, __bf16>), dim3(blocks), dim3(256), 0, st, (const float*)x, (__bf16*)y, (long long)n);
    else if (xdt == VITK_BF16 && ydt == VITK_F32) hipLaunchKernelGGL((cast_kernel<__bf16, float>), dim3(blocks), dim3(256), 0, st, (const __bf16*)x, (float*)y, (long long)n);
    else if (xdt == VITK_BF16 && ydt == VITK_BF16) hipLaunchKernelGGL((cast_kernel<__bf16, __bf16>), dim3(blocks), dim3(256), 0, st, (const __bf16*)x, (__bf16*)y, (long long)n);
    else VITK_FAIL(VITK_E_DTYPE, "cast: bad dtype");
    VITK_CHECK_LAUNCH("cast");
    return 0;
}

extern "C" int vitk_cast_many(const void* const* src, void* const* dst, const int64_t* numel, int64_t count, int xdt, int ydt, void* stream) {
    if (count <= 0) return 0;
    if (!src || !dst || !numel) VITK_FAIL(VITK_E_ARG, "cast_many: null table");
    if (!((xdt == VITK_F32 || xdt == VITK_BF16) && (ydt == VITK_F32 || ydt == VITK_BF16))) VITK_FAIL(VITK_E_DTYPE, "cast_many: bad dtype");
    hipStream_t st = (hipStream_t)stream;
    int64_t t = 0;
    while (t < count) {
        CastMany a;
        a.count = 0; a.blk0[0] = 0;
        while (t < count && a.count < CM_MAX) {
            const int64_t n = numel[t];
            if (n < 0 || (n > 0 && (!src[t] || !dst[t]))) VITK_FAIL(VITK_E_ARG, "cast_many: null pointer / negative size in the table");
            if (n > 0) {
                // 16-byte vector accesses on the float32 side, 8-byte on the 16-bit side
                if (((uintptr_t)src[t] & (xdt == VITK_F32 ? 15 : 7)) || ((uintptr_t)dst[t] & (ydt == VITK_F32 ? 15 : 7))) VITK_FAIL(VITK_E_ARG, "cast_many: misaligned tensor");
                const long long blocks = (n + CM_PER_BLOCK - 1) / CM_PER_BLOCK;
                if (a.blk0[a.count] + blocks > 0x3fffffff) { if (a.count == 0) VITK_FAIL(VITK_E_SHAPE, "cast_many: tensor too large"); break; }
                a.src[a.count] = src[t]; a.dst[a.count] = dst[t]; a.n[a.count] = n;
                a.blk0[a.count + 1] = a.blk0[a.count] + (int)blocks;
                ++a.count;
            }
            ++t;
        }
        if (a.count == 0) continue;
        const dim3 grid((unsigned)a.blk0[a.count]);
        if (xdt == VITK_F32 && ydt == VITK_BF16) hipLaunchKernelGGL((cast_many_kernel<float, __bf16>), grid, dim3(256), 0, st, a);
        else if (xdt == VITK_BF16 && ydt == VITK_F32) hipLaunchKernelGGL((cast_many_kernel<__bf16, float>), grid, dim3(256), 0, st, a);
        else if (xdt == VITK_F32) hipLaunchKernelGGL((cast_many_kernel<float, float>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((cast_many_kernel<__bf16, __bf16>), grid, dim3(256), 0, st, a);
        VITK_CHECK_LAUNCH("cast_many");
    }
    return 0;
}

extern "C" int vitk_adam_step(void* param, const void* grad, int dt, float* exp_avg, float* exp_avg_sq, float* master, int64_t n,
                              float lr, float beta1, float beta2, float eps, float weight_decay, int decoupled, int64_t step,
                              float grad_scale, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq) VITK_FAIL(VITK_E_ARG, "adam_step: null pointer");
    if (n <= 0) return 0;
    if (step < 1) VITK_FAIL(VITK_E_ARG, "adam_step: step counts from 1 (got %lld)", (long long)step);
    if (!aligned16(param) || !aligned16(grad) || !aligned16(exp_avg) || !aligned16(exp_avg_sq) || (master && !aligned16(master)))
        VITK_FAIL(VITK_E_ALIGN, "adam_step: pointers must be 16-byte aligned");
    const double c1 = 1.0 - pow((double)beta1, (double)step), c2 = 1.0 - pow((double)beta2, (double)step);
    const float inv_c1 = (float)(1.0 / c1), inv_sqrt_c2 = (float)(1.0 / sqrt(c2));
    hipStream_t st = (hipStream_t)stream;
    const unsigned blocks = ew_blocks((n + 3) / 4);
    VITK_DISPATCH_DT(dt, T, hipLaunchKernelGGL((adam_kernel<T>), dim3(blocks), dim3(256), 0, st, (T*)param, (const T*)grad, exp_avg,
                                                exp_avg_sq, master, (long long)n, lr, beta1, beta2, eps, weight_decay, decoupled, inv_c1,
                                                inv_sqrt_c2, grad_scale));
    VITK_CHECK_LAUNCH("adam_step");
    return 0;
}

extern "C" int vitk_fp8_amax_scale(const void* x, int dt, int64_t n, float* scale2, void* stream) {
    if (!x || !scale2) VITK_FAIL(VITK_E_ARG, "fp8_amax_scale: null pointer");
    if (n <= 0) VITK_FAIL(VITK_E_SHAPE, "fp8_amax_scale: empty tensor");
    hipStream_t st = (hipStream_t)stream;
    unsigned* bits = reinterpret_cast<unsigned*>(scale2) + 1;          // scale2[1] doubles as the amax scratch word
    if (hipMemsetAsync(bits, 0, sizeof(unsigned), st) != hipSuccess) VITK_FAIL(1, "fp8_amax_scale: memset failed");
    unsigned blocks = ew_blocks((n + 3) / 4); if (blocks > 1024) blocks = 1024;
    VITK_DISPATCH_DT(dt, T, hipLaunchKernelGGL((absmax_kernel<T>), dim3(blocks), dim3(256), 0, st, (const T*)x, (long long)n, bits));
    hipLaunchKernelGGL(fp8_scale_kernel, dim3(1), dim3(1), 0, st, bits, scale2);
    VITK_CHECK_LAUNCH("fp8_amax_scale");
    return 0;
}

extern "C" int vitk_quantize_fp8(const void* x, int dt, void* out, int64_t n, const float* scale_dev, float scale_host, void* stream) {
    if (!x || !out) VITK_FAIL(VITK_E_ARG, "quantize_fp8: null pointer");
    if (n <= 0 || (n & 3)) VITK_FAIL(VITK_E_SHAPE, "quantize_fp8: n must be a positive multiple of 4");
    if (!aligned16(x) || !aligned16(out)) VITK_FAIL(VITK_E_ALIGN, "quantize_fp8: 16-byte aligned pointers required");
    hipStream_t st = (hipStream_t)stream;
    const unsigned blocks = ew_blocks(n / 4);
    VITK_DISPATCH_DT(dt, T, hipLaunchKernelGGL((quantize_fp8_kernel<T>), dim3(blocks), dim3(256), 0, st, (const T*)x, (unsigned*)out,
                                                (long long)(n / 4), scale_dev, scale_host));
    VITK_CHECK_LAUNCH("quantize_fp8");
    return 0;
}

// ---- test hook: hold `ncus` CUs for `ms` milliseconds (one 1024-thread, 160 KiB-LDS workgroup per CU, sleeping on the 100 MHz
// real-time counter) -- stands in for a collective's resident kernel when co-scheduling is measured on ONE GPU (tools/cu_contention.py)
namespace {
__global__ __launch_bounds__(1024) void occupy_kernel(long long ticks) {
    extern __shared__ char hold[];
    if (threadIdx.x == 0) hold[0] = 1;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(127);
}
}  // namespace
extern "C" int vitk_test_occupy_cus(int ncus, float ms, void* stream) {
    if (ncus < 1 || ncus > 256 || !(ms > 0.f) || ms > 1000.f) VITK_FAIL(VITK_E_ARG, "test_occupy_cus: 1..256 CUs, 0 < ms <= 1000");
    static const int rc__ = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (rc__ != 0) VITK_FAIL(rc__, "test_occupy_cus: cannot enable 160 KiB of LDS");
    hipLaunchKernelGGL(occupy_kernel, dim3((unsigned)ncus), dim3(1024), 160 * 1024, (hipStream_t)stream, (long long)(ms * 1e5f));
    VITK_CHECK_LAUNCH("test_occupy_cus");
    return 0;
}

extern "C" int vitk_dropout_keep(uint8_t* keep, int64_t rows, int64_t cols, float p, uint32_t seed, void* stream) {
    if (!keep) VITK_FAIL(VITK_E_ARG, "dropout_keep: null pointer");
    if (rows <= 0 || cols <= 0 || rows > 0xffffffffLL || cols > 0x7fffffffLL) VITK_FAIL(VITK_E_SHAPE, "dropout_keep: bad extents");
    hipLaunchKernelGGL(dropout_keep_kernel, dim3(ew_blocks(rows * cols)), dim3(256), 0, (hipStream_t)stream, keep, (long long)rows,
                       (long long)cols, drop_thresh(p), (unsigned)seed);
    VITK_CHECK_LAUNCH("dropout_keep");
    return 0;
}

extern "C" int vitk_fp8_update_scales(uint32_t* amax64, float* scales2, int64_t nslots, void* stream) {
    if (!amax64 || !scales2) VITK_FAIL(VITK_E_ARG, "fp8_update_scales: null pointer");
    if (nslots <= 0) return 0;
    hipLaunchKernelGGL(fp8_update_scales_kernel, dim3((unsigned)nslots), dim3(64), 0, (hipStream_t)stream, (unsigned*)amax64, scales2, (int)nslots);
    VITK_CHECK_LAUNCH("fp8_update_scales");
    return 0;
}

extern "C" int vitk_fp8_update_scales_fmt(uint32_t* amax64, float* scales2, int64_t nslots, const float* fmax, void* stream) {
    if (!amax64 || !scales2) VITK_FAIL(VITK_E_ARG, "fp8_update_scales_fmt: null pointer");
    if (nslots <= 0) return 0;
    hipLaunchKernelGGL(fp8_update_scales_fmt_kernel, dim3((unsigned)nslots), dim3(64), 0, (hipStream_t)stream, (unsigned*)amax64, scales2,
                       (int)nslots, fmax);
    VITK_CHECK_LAUNCH("fp8_update_scales_fmt");
    return 0;
}

extern "C" int vitk_quantize_fp8_delayed(const void* x, int dt, void* out8, int64_t n, const float* scale2, uint32_t* amax64, int fmt,
                                         void* stream) {
    if (!x || (!out8 && !amax64)) VITK_FAIL(VITK_E_ARG, "quantize_fp8_delayed: null input, or neither an output nor an amax record");
    if (out8 && !scale2) VITK_FAIL(VITK_E_ARG, "quantize_fp8_delayed: an fp8 output needs its scale");
    if (fmt != 0 && fmt != 1) VITK_FAIL(VITK_E_ARG, "quantize_fp8_delayed: fmt is 0 (e4m3) or 1 (e5m2), got %d", fmt);
    if (n <= 0 || (n & 3)) VITK_FAIL(VITK_E_SHAPE, "quantize_fp8_delayed: n must be a positive multiple of 4");
    if (!aligned16(x) || (out8 && !aligned16(out8))) VITK_FAIL(VITK_E_ALIGN, "quantize_fp8_delayed: 16-byte aligned pointers required");
    hipStream_t st = (hipStream_t)stream;
    unsigned blocks = ew_blocks(n / 4); if (blocks > 4096) blocks = 4096;
    if (fmt == 1) {
        VITK_DISPATCH_DT(dt, T, hipLaunchKernelGGL((quantize_fp8_delayed_kernel<T, true>), dim3(blocks), dim3(256), 0, st, (const T*)x,
                                                    (unsigned*)out8, (long long)(n / 4), scale2, (unsigned*)amax64));
    } else {
        VITK_DISPATCH_DT(dt, T, hipLaunchKernelGGL((quantize_fp8_delayed_kernel<T, false>), dim3(blocks), dim3(256), 0, st, (const T*)x,
                                                    (unsigned*)out8, (long long)(n / 4), scale2, (unsigned*)amax64));
    }
    VITK_CHECK_LAUNCH("quantize_fp8_delayed");
    return 0;
}

extern "C" int vitk_write_cls_rows(void* x, int xdt, const void* cls, const void* pos, int pdt, int64_t B, int64_t N, int64_t D,
                                   int64_t ncls, void* stream) {
    if (ncls == 0) return 0;
    if (!x || !cls || !pos) VITK_FAIL(VITK_E_ARG, "write_cls_rows: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const unsigned blocks = ew_blocks(B * ncls * D);
    if (xdt == VITK_F32 && pdt == VITK_F32) hipLaunchKernelGGL((write_cls_kernel<float, float>), dim3(blocks), dim3(256), 0, st, (float*)x, (const float*)cls, (const float*)pos, (long long)B, (long long)N, (int)D, (int)ncls);
    else if (xdt == VITK_F32 && pdt == VITK_BF16) hipLaunchKernelGGL((write_cls_kernel<float, __bf16>), dim3(blocks), dim3(256), 0, st, (float*)x, (const __bf16*)cls, (const __bf16*)pos, (long long)B, (long long)N, (int)D, (int)ncls);
    else if (xdt == VITK_BF16 && pdt == VITK_BF16) hipLaunchKernelGGL((write_cls_kernel<__bf16, __bf16>), dim3(blocks), dim3(256), 0, st, (__bf16*)x, (const __bf16*)cls, (const __bf16*)pos, (long long)B, (long long)N, (int)D, (int)ncls);
    else VITK_FAIL(VITK_E_DTYPE, "write_cls_rows: bad dtype combination");
    VITK_CHECK_LAUNCH("write_cls_rows");
    return 0;
}

extern "C" int vitk_mean_pool_fwd(const void* x, int xdt, void* out, int odt, int64_t B, int64_t N, int64_t D, void* stream) {
    if (!x || !out) VITK_FAIL(VITK_E_ARG, "mean_pool_fwd: null pointer");
    if (B <= 0 || N <= 0 || D <= 0 || (D & 3)) VITK_FAIL(VITK_E_SHAPE, "mean_pool_fwd: D %% 4 != 0");
    hipStream_t st = (hipStream_t)stream;
    const unsigned blocks = ew_blocks(B * D / 4);
    if (xdt == VITK_F32 && odt == VITK_F32) hipLaunchKernelGGL((mean_pool_fwd_kernel<float, float>), dim3(blocks), dim3(256), 0, st, (const float*)x, (float*)out, (long long)B, (int)N, (int)(D / 4));
    else if (xdt == VITK_BF16 && odt == VITK_BF16) hipLaunchKernelGGL((mean_pool_fwd_kernel<__bf16, __bf16>), dim3(blocks), dim3(256), 0, st, (const __bf16*)x, (__bf16*)out, (long long)B, (int)N, (int)(D / 4));
    else if (xdt == VITK_F32 && odt == VITK_BF16) hipLaunchKernelGGL((mean_pool_fwd_kernel<float, __bf16>), dim3(blocks), dim3(256), 0, st, (const float*)x, (__bf16*)out, (long long)B, (int)N, (int)(D / 4));
    else VITK_FAIL(VITK_E_DTYPE, "mean_pool_fwd: bad dtype combination");
    VITK_CHECK_LAUNCH("mean_pool_fwd");
    return 0;
}
extern "C" int vitk_mean_pool_bwd(const void* dout, int ddt, void* dx, int xdt, int64_t B, int64_t N, int64_t D, void* stream) {
    if (!dout || !dx) VITK_FAIL(VITK_E_ARG, "mean_pool_bwd: null pointer");
    if (B <= 0 || N <= 0 || D <= 0 || (D & 3)) VITK_FAIL(VITK_E_SHAPE, "mean_pool_bwd: D %% 4 != 0");
    hipStream_t st = (hipStream_t)stream;
    const unsigned blocks = ew_blocks(B * N * D / 4);
    if (ddt == VITK_F32 && xdt == VITK_F32) hipLaunchKernelGGL((mean_pool_bwd_kernel<float, float>), dim3(blocks), dim3(256), 0, st, (const float*)dout, (float*)dx, (long long)B, (int)N, (int)(D / 4));
    else if (ddt == VITK_BF16 && xdt == VITK_BF16) hipLaunchKernelGGL((mean_pool_bwd_kernel<__bf16, __bf16>), dim3(blocks), dim3(256), 0, st, (const __bf16*)dout, (__bf16*)dx, (long long)B, (int)N, (int)(D / 4));
    else if (ddt == VITK_BF16 && xdt == VITK_F32) hipLaunchKernelGGL((mean_pool_bwd_kernel<__bf16, float>), dim3(blocks), dim3(256), 0, st, (const __bf16*)dout, (float*)dx, (long long)B, (int)N, (int)(D / 4));
    else VITK_FAIL(VITK_E_DTYPE, "mean_pool_bwd: bad dtype combination");
    VITK_CHECK_LAUNCH("mean_pool_bwd");
    return 0;
}

extern "C" int vitk_dropout_fwd(const void* x, void* y, uint8_t* mask, int dt, int64_t n, float p, uint64_t seed, uint64_t offset,
                                void* stream) {
    if (!x || !y || !mask) VITK_FAIL(VITK_E_ARG, "dropout_fwd: null pointer");
    if (n <= 0 || !(p >= 0.f && p < 1.f)) VITK_FAIL(VITK_E_SHAPE, "dropout_fwd: need n > 0 and 0 <= p < 1");
    VITK_DISPATCH_DT(dt, T, hipLaunchKernelGGL((dropout_fwd_kernel<T>), dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream,
                                                (const T*)x, (T*)y, mask, (long long)n, p, (unsigned long long)seed,
                                                (unsigned long long)offset));
    VITK_CHECK_LAUNCH("dropout_fwd");
    return 0;
}
extern "C" int vitk_dropout_bwd(const void* dy, const uint8_t* mask, void* dx, int dt, int64_t n, float p, void* stream) {
    if (!dy || !dx || !mask) VITK_FAIL(VITK_E_ARG, "dropout_bwd: null pointer");
    if (n <= 0 || !(p >= 0.f && p < 1.f)) VITK_FAIL(VITK_E_SHAPE, "dropout_bwd: need n > 0 and 0 <= p < 1");
    VITK_DISPATCH_DT(dt, T, hipLaunchKernelGGL((dropout_bwd_kernel<T>), dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream,
                                                (const T*)dy, mask, (T*)dx, (long long)n, p));
    VITK_CHECK_LAUNCH("dropout_bwd");
    return 0;
}

extern "C" int vitk_transpose(const void* in, void* out, int dt, int64_t rows, int64_t cols, void* stream) {
    if (!in || !out) VITK_FAIL(VITK_E_ARG, "transpose: null pointer");
    if (rows <= 0 || cols <= 0 || (rows + 31) / 32 > 65535) VITK_FAIL(VITK_E_SHAPE, "transpose: bad shape");
    const dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
    VITK_DISPATCH_DT(dt, T, hipLaunchKernelGGL((transpose_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)in,
                                                (T*)out, (int)rows, (int)cols));
    VITK_CHECK_LAUNCH("transpose");
    return 0;
}

extern "C" int vitk_patchify_cpp(const void* img, void* out, int dt, int64_t C, int64_t H, int64_t W, int64_t p, int64_t row0,
                                 int64_t ld, void* stream) {
    if (!img || !out) VITK_FAIL(VITK_E_ARG, "patchify_cpp: null pointer");
    if (C <= 0 || p <= 0 || H <= 0 || W <= 0 || H % p || W % p) VITK_FAIL(VITK_E_SHAPE, "patchify_cpp: image not divisible by patch");
    const long long rows = (H / p) * (W / p);
    VITK_DISPATCH_DT(dt, T, hipLaunchKernelGGL((patchify_cpp_kernel<T>), dim3(ew_blocks(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                                                (const T*)img, (T*)out, (int)C, (int)H, (int)W, (int)p, (long long)row0, (long long)ld));
    VITK_CHECK_LAUNCH("patchify_cpp");
    return 0;
}

extern "C" int vitk_unpatchify_cpp(const void* dpatch, void* dimg, int dt, int64_t C, int64_t H, int64_t W, int64_t p, int64_t row0,
                                   int64_t ld, void* stream) {
    if (!dpatch || !dimg) VITK_FAIL(VITK_E_ARG, "unpatchify_cpp: null pointer");
    if (C <= 0 || p <= 0 || H <= 0 || W <= 0 || H % p || W % p) VITK_FAIL(VITK_E_SHAPE, "unpatchify_cpp: image not divisible by patch");
    const long long rows = (H / p) * (W / p);
    VITK_DISPATCH_DT(dt, T, hipLaunchKernelGGL((unpatchify_cpp_kernel<T>), dim3(ew_blocks(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                                                (const T*)dpatch, (T*)dimg, (int)C, (int)H, (int)W, (int)p, (long long)row0, (long long)ld));
    VITK_CHECK_LAUNCH("unpatchify_cpp");
    return 0;
}

extern "C" int vitk_gather_add2(const void* x, const void* A, const int32_t* ia, const void* B, const int32_t* ib, void* out, int dt,
                                int64_t T, int64_t D, void* stream) {
    if (!x || !A || !ia || !B || !ib || !out) VITK_FAIL(VITK_E_ARG, "gather_add2: null pointer");
    if (T <= 0 || D <= 0 || (D & 3)) VITK_FAIL(VITK_E_SHAPE, "gather_add2: D %% 4 != 0");
    VITK_DISPATCH_DT(dt, Tt, hipLaunchKernelGGL((gather_add2_kernel<Tt>), dim3(ew_blocks(T * D / 4)), dim3(256), 0, (hipStream_t)stream,
                                                 (const Tt*)x, (const Tt*)A, ia, (const Tt*)B, ib, (Tt*)out, (long long)T, (int)(D / 4)));
    VITK_CHECK_LAUNCH("gather_add2");
    return 0;
}

extern "C" int vitk_csr_rowsum(const void* g, int gdt, const int32_t* ptr, const int32_t* rows, void* out, int odt, int64_t nseg,
                               int64_t D, void* stream) {
    if (!g || !ptr || !rows || !out) VITK_FAIL(VITK_E_ARG, "csr_rowsum: null pointer");
    if (nseg <= 0 || D <= 0 || (D & 3) || nseg > 0x7fffffff) VITK_FAIL(VITK_E_SHAPE, "csr_rowsum: D %% 4 != 0");
    hipStream_t st = (hipStream_t)stream;
    const int D4 = (int)(D / 4);
    if (gdt == VITK_F32 && odt == VITK_F32) hipLaunchKernelGGL((csr_rowsum_kernel<float, float>), dim3((unsigned)nseg), dim3(256), 0, st, (const float*)g, ptr, rows, (float*)out, D4);
    else if (gdt == VITK_BF16 && odt == VITK_BF16) hipLaunchKernelGGL((csr_rowsum_kernel<__bf16, __bf16>), dim3((unsigned)nseg), dim3(256), 0, st, (const __bf16*)g, ptr, rows, (__bf16*)out, D4);
    else if (gdt == VITK_F32 && odt == VITK_BF16) hipLaunchKernelGGL((csr_rowsum_kernel<float, __bf16>), dim3((unsigned)nseg), dim3(256), 0, st, (const float*)g, ptr, rows, (__bf16*)out, D4);
    else if (gdt == VITK_BF16 && odt == VITK_F32) hipLaunchKernelGGL((csr_rowsum_kernel<__bf16, float>), dim3((unsigned)nseg), dim3(256), 0, st, (const __bf16*)g, ptr, rows, (float*)out, D4);
    else VITK_FAIL(VITK_E_DTYPE, "csr_rowsum: bad dtype");
    VITK_CHECK_LAUNCH("csr_rowsum");
    return 0;
}

extern "C" int vitk_copy_cols(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int dt, int64_t rows, int64_t cols_copy,
                              int64_t cols_dst, void* stream) {
    if (!src || !dst) VITK_FAIL(VITK_E_ARG, "copy_cols: null pointer");
    if (rows <= 0 || cols_copy < 0 || cols_dst <= 0 || cols_copy > cols_dst || cols_dst > ld_dst || cols_copy > ld_src)
        VITK_FAIL(VITK_E_SHAPE, "copy_cols: bad extents");
    VITK_DISPATCH_DT(dt, T, hipLaunchKernelGGL((copy_cols_kernel<T>), dim3(ew_blocks(rows * cols_dst)), dim3(256), 0, (hipStream_t)stream,
                                                (const T*)src, (long long)ld_src, (T*)dst, (long long)ld_dst, (long long)rows, (int)cols_copy, (int)cols_dst));
    VITK_CHECK_LAUNCH("copy_cols");
    return 0;
}
