// common.h -- shared device/host helpers for libvitk (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/vitk.h"

// libvitk_f16.so is built from these same sources with -DVITK_HALF_IS_F16: every "bf16" below then IS IEEE binary16
// (_Float16) -- 16-bit storage, the same 8-element fragments, v_mfma_f32_16x16x32_f16 instead of ..._bf16, f32
// accumulation everywhere -- and dtype tag 1 means "this library's 16-bit type" (vitk_half_type() says which).
// The switch sits after the HIP headers so that their own bf16 helpers are left alone.
#ifdef VITK_HALF_IS_F16
#define __bf16 _Float16
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16 __builtin_amdgcn_mfma_f32_16x16x32_f16
#endif

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define WAVE 64

// ---- error plumbing (host) -------------------------------------------------------------
void vitk_set_error(const char* fmt, ...);

// ---- runtime switches (host) -------------------------------------------------------------
// vitk_switch(NAME): the handful of environment switches the product build reads (README "Switches"): kernel-selection hooks the test
// suite and tools/nt_probe use to hold two production kernels against each other (VITK_NT_W128, VITK_ATTN_PIPE, VITK_ATTN_R_*,
// VITK_NTP_STATIC / _DYNAMIC, VITK_TN_PAIR).  vitk_exp(NAME): experiment knobs (tile orders, cost-model constants, debug stamps, ablations) --
// compiled out of the product build; `VITK_BUILD_EXPERIMENTS=1 python -m vit_pytorch_amd._build` brings them back for the tools/ scripts.
const char* vitk_switch(const char* name);
#ifdef VITK_EXPERIMENTS
#define vitk_exp(name) vitk_switch(name)
#else
#define vitk_exp(name) (static_cast<const char*>(nullptr))
#endif
#define VITK_FAIL(code, ...) do { vitk_set_error(__VA_ARGS__); return (code); } while (0)
#define VITK_CHECK_LAUNCH(name) do { hipError_t e__ = hipGetLastError(); \
    if (e__ != hipSuccess) { vitk_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); return (int)e__; } } while (0)
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline bool aligned8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; }

// ---- bf16 <-> f32 (device) ---------------------------------------------------------------
// round-to-nearest-even, NaN preserved: what torch's .to(bfloat16) does.
__device__ __forceinline__ float bf2f(__bf16 v) { return (float)v; }
__device__ __forceinline__ __bf16 f2bf(float f) { return (__bf16)f; }

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<__bf16>(__bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 from_f32<__bf16>(float v) { return (__bf16)v; }

// 4-element vector load/store as f32x4 (16 B for f32, 8 B for bf16). p must be aligned to the
// vector size.
template <typename T> __device__ __forceinline__ f32x4 load4(const T* p);
template <> __device__ __forceinline__ f32x4 load4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <> __device__ __forceinline__ f32x4 load4<__bf16>(const __bf16* p) {
    bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
template <typename T> __device__ __forceinline__ void store4(T* p, f32x4 v);
template <> __device__ __forceinline__ void store4<float>(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> __device__ __forceinline__ void store4<__bf16>(__bf16* p, f32x4 v) {
    bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    *reinterpret_cast<bf16x4*>(p) = o;
}

// ---- wave reductions (wave = 64) --------------------------------------------------------
// All 64 lanes receive the result.  Rotations inside each row of 16 lanes are DPP modifiers and the two cross-row
// levels are v_permlane16_swap / v_permlane32_swap: pure VALU, no ds_bpermute round trips through the LDS crossbar
// (6 dependent ~100-cycle hops per reduction with __shfl_xor) and no s_waitcnt lgkmcnt in the middle of a row.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f32<0x128>(v);   // row_ror:8
    v += dpp_f32<0x124>(v);   // row_ror:4
    v += dpp_f32<0x122>(v);   // row_ror:2
    v += dpp_f32<0x121>(v);   // row_ror:1
    unsigned u = __builtin_bit_cast(unsigned, v);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = __builtin_bit_cast(float, (unsigned)a[0]) + __builtin_bit_cast(float, (unsigned)a[1]);
    u = __builtin_bit_cast(unsigned, v);
    auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)b[0]) + __builtin_bit_cast(float, (unsigned)b[1]);
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f32<0x128>(v));
    v = fmaxf(v, dpp_f32<0x124>(v));
    v = fmaxf(v, dpp_f32<0x122>(v));
    v = fmaxf(v, dpp_f32<0x121>(v));
    unsigned u = __builtin_bit_cast(unsigned, v);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = fmaxf(__builtin_bit_cast(float, (unsigned)a[0]), __builtin_bit_cast(float, (unsigned)a[1]));
    u = __builtin_bit_cast(unsigned, v);
    auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)b[0]), __builtin_bit_cast(float, (unsigned)b[1]));
}

// ---- math ------------------------------------------------------------------------------
// Exact-erf GELU (nn.GELU() default, vit.py:21).  Phi(x) = 0.5*erfc(-x/sqrt2) is evaluated with the
// Abramowitz-Stegun 7.1.26 rational form of erfc (|abs err| <= 1.5e-7, i.e. f32 round-off class; no
// cancellation in the negative tail because erfc is formed directly): 1 rcp + 1 exp2 + ~8 FMA instead of
// the ~40-instruction libm erff.  The same exp(-x^2/2) serves the Gaussian density in the derivative.
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& pdf) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    const float e = __builtin_amdgcn_exp2f(-1.44269504088896340736f * z * z);   // exp(-z^2) = exp(-x^2/2)
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float half_erfc = 0.5f * p * t * e;           // 0.5 * erfc(|x|/sqrt2)
    cdf = x < 0.f ? half_erfc : 1.0f - half_erfc;
    pdf = e * 0.39894228040143267794f;
}
__device__ __forceinline__ float gelu_erf(float x) { float c, p; gelu_parts(x, c, p); return x * c; }
__device__ __forceinline__ float gelu_erf_grad(float x) { float c, p; gelu_parts(x, c, p); return fmaf(x, p, c); }

// Cheaper Phi(x) for the bf16 GEMM epilogues, two elements at a time so the FMAs issue as v_pk_fma_f32
// (the epilogue of a 256 x 256 tile evaluates 65,536 GELUs on the VALU while the matrix cores idle; the
// rational form above costs ~26 issue slots per element, this one ~6).  Odd polynomial of degree 17 in x
// on [-4.5, 4.5] (Chebyshev fit, evaluated in x^2 by Horner), input clamped to the interval:
// |abs err| <= 2.2e-5 everywhere, i.e. ~1/50 of the bf16 rounding step of the result it feeds.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }
__device__ __forceinline__ f32x2 phi_poly2(f32x2 x) {
    const f32x2 xc = {__builtin_amdgcn_fmed3f(x[0], -4.5f, 4.5f), __builtin_amdgcn_fmed3f(x[1], -4.5f, 4.5f)};
    const f32x2 u = xc * xc;
    f32x2 q = __builtin_elementwise_fma(u, splat2(3.619783835e-11f), splat2(-3.842468662e-09f));
    q = __builtin_elementwise_fma(q, u, splat2(1.789582063e-07f));
    q = __builtin_elementwise_fma(q, u, splat2(-4.853476327e-06f));
    q = __builtin_elementwise_fma(q, u, splat2(8.614045158e-05f));
    q = __builtin_elementwise_fma(q, u, splat2(-1.069849927e-03f));
    q = __builtin_elementwise_fma(q, u, splat2(9.707349039e-03f));
    q = __builtin_elementwise_fma(q, u, splat2(-6.620850869e-02f));
    q = __builtin_elementwise_fma(q, u, splat2(3.988530737e-01f));
    return __builtin_elementwise_fma(xc, q, splat2(0.5f));
}
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) { return x * phi_poly2(x); }
// d/dx [x Phi(x)] = Phi(x) + x phi(x); the density keeps its exp2 (no stable low-degree polynomial over the range)
__device__ __forceinline__ f32x2 gelu_grad_fast2(f32x2 x) {
    const f32x2 w = x * x * splat2(-0.72134752044448170368f);
    const f32x2 e = {__builtin_amdgcn_exp2f(w[0]), __builtin_amdgcn_exp2f(w[1])};
    return __builtin_elementwise_fma(x * splat2(0.39894228040143267794f), e, phi_poly2(x));
}

// ---- fp8 (e4m3) side output of a producer kernel (delayed scaling) ---------------------------------------------------
// A producer that already holds the values in registers (LayerNorm forward, the GELU epilogue) can emit an e4m3 copy for the
// next GEMM: p = destination (1 byte / element, same row-major layout; null = no copy), scale = device pointer to the
// quantisation scale decided BEFORE this step (448 / amax of the previous step), amax = 64 device words that collect
// max|value| of THIS step as float bit patterns (null = do not record); vitk_fp8_update_scales() turns them into the next
// step's scales.  No pass over the tensor is added and the step never waits for its own statistics.
struct F8Out { unsigned char* p; const float* scale; unsigned* amax; };
__device__ __forceinline__ unsigned pack_fp8x4(f32x4 v, float sc) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e] * sc, -448.f, 448.f);
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w, true);
    return (unsigned)w;
}
// the same for OCP e5m2 ("bf8"; gradients): clamp at +-57344, v_cvt_pk_bf8_f32
__device__ __forceinline__ unsigned pack_bf8x4(f32x4 v, float sc) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e] * sc, -57344.f, 57344.f);
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_bf8_f32(v[0], v[1], w, false);
    w = __builtin_amdgcn_cvt_pk_bf8_f32(v[2], v[3], w, true);
    return (unsigned)w;
}
__device__ __forceinline__ float absmax4(f32x4 v) { return fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))); }

// ---- counter-based dropout decisions for the fused paths ------------------------------------------------------------
// keep(row, col) = hash32(hash32(row ^ seed) + col) >= thresh, thresh = p * 2^32.  Stateless: the forward kernel, the
// backward kernels and vitk_dropout_keep() (the test hook) regenerate the same decision from (seed, row, col), so no
// mask tensor is stored or read.  hash32 = the "lowbias32" integer finaliser (2 multiplies, full avalanche).
__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x21f0aaadu;
    x ^= x >> 15; x *= 0x735a2d97u;
    x ^= x >> 15;
    return x;
}
__device__ __forceinline__ unsigned drop_row(unsigned row, unsigned seed) { return hash32(row ^ seed); }
__device__ __forceinline__ bool drop_keep(unsigned hrow, unsigned col, unsigned thresh) { return hash32(hrow + col) >= thresh; }
static inline unsigned drop_thresh(float p) {
    const double t = (double)p * 4294967296.0;
    return t <= 0.0 ? 0u : (t >= 4294967295.0 ? 4294967295u : (unsigned)t);
}

// ---- row map (see vitk.h) ---------------------------------------------------------------
struct RowMap { long long group, gstride, offset; };
__device__ __forceinline__ long long map_row(const RowMap& m, long long r) {
    if (m.group <= 0) return r;
    return (r / m.group) * m.gstride + (r % m.group) + m.offset;
}
static inline RowMap to_map(vitk_rowmap m) { return RowMap{(long long)m.group, (long long)m.gstride, (long long)m.offset}; }

// dtype dispatch helpers (host)
#define VITK_DISPATCH_DT(dt, T, ...) \
    if ((dt) == VITK_F32) { using T = float; __VA_ARGS__; } \
    else if ((dt) == VITK_BF16) { using T = __bf16; __VA_ARGS__; } \
    else VITK_FAIL(VITK_E_DTYPE, "bad dtype tag %d", (int)(dt))
