// gemm_nt_epi.h -- epilogue helpers shared by the NT GEMM kernels (gemm_nt_persist.hip: 8 waves, 128 x 64 wave tiles;
// gemm_nt_w128.hip: 4 waves, 128 x 128 wave tiles): uncounted asm loads with exact-count waits, the packed GELU forms, the
// lane-pair exchange that turns a lane's 4 columns of 2 rows into 8 columns of one row (full 128-byte lines per store).
#pragma once
#include "common.h"

namespace {

// ---- epilogue operand loads the compiler does not count (interior tiles) ---------------------------------------------------
// hipcc waits vmcnt(0) for an ordinary VGPR-destination load whenever an LDS-DMA is in flight -- and in this kernel one always is
// (the next tile's first K-steps).  CDNA4's vmcnt counts stores too and retires in order, so every use of a residual row (f32
// residual epilogue) or of a saved pre-activation row (GELU' epilogue) drained the ring AND waited for the completion of every
// store issued so far: a write round trip per fragment row, 4-8 of them per tile [the ISA showed vmcnt(0) / vmcnt(1) in front of
// every second row; FF2 + residual: ~26 us of epilogue per 256 x 256 tile against ~9 us of store issue].  Issued here as asm
// and waited for by an EXACT count: D fragment rows are in flight, and the stores issued after a row's loads may stay in flight.
// (Form (ii) of the guide's 5.7: "=v" loads, then a wait statement naming every destination "+v"; the epilogue is straight-line
// code, so no destination is loop-carried; tools/asm_inflight_audit.py checks the .s for compiler accesses in between.)
#ifndef Q_EPI_ASM_RESID
#define Q_EPI_ASM_RESID 1        // 0: compiler-counted residual loads (A/B builds)
#endif
#ifndef Q_EPI_ASM_PRE
#define Q_EPI_ASM_PRE 1          // 0: compiler-counted pre-activation loads (A/B builds)
#endif
#ifndef Q_EPI_DEPTH_RESID
#define Q_EPI_DEPTH_RESID 3      // fragment rows of the residual in flight (16 VGPRs each)
#endif
#ifndef Q_EPI_DEPTH_PRE
#define Q_EPI_DEPTH_PRE 2        // fragment rows of the saved pre-activation in flight (8 VGPRs each)
#endif
__device__ __forceinline__ void q_gload_f32x4(f32x4& d, const float* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ void q_gload_bf16x8(bf16x8& d, const __bf16* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
// ops issued between the loads of fragment row f and their wait: PER_F loads + PER_F stores per row, D rows of loads in flight
__host__ __device__ constexpr int q_epi_younger(int f, int fmw, int d, int per_f) {
    const int last = f + d - 1 < fmw - 1 ? f + d - 1 : fmw - 1;
    return per_f * ((last - f) + (f < d ? f : d - 1));
}
#define Q_WAIT_CASE(n) else if constexpr (N_ == n) asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "memory")
template <int N_, typename T> __device__ __forceinline__ void q_wait_regs4(T& a, T& b, T& c, T& d) {
    if constexpr (N_ == 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "memory");
    Q_WAIT_CASE(4); Q_WAIT_CASE(6); Q_WAIT_CASE(8); Q_WAIT_CASE(10); Q_WAIT_CASE(12); Q_WAIT_CASE(16); Q_WAIT_CASE(20); Q_WAIT_CASE(24);
    Q_WAIT_CASE(28); Q_WAIT_CASE(32); Q_WAIT_CASE(36); Q_WAIT_CASE(40); Q_WAIT_CASE(44); Q_WAIT_CASE(48); Q_WAIT_CASE(52); Q_WAIT_CASE(56); Q_WAIT_CASE(60);
    else static_assert(N_ < 0, "unsupported vmcnt");
}
#define Q_WAIT2_CASE(n) else if constexpr (N_ == n) asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(a), "+v"(b) :: "memory")
template <int N_, typename T> __device__ __forceinline__ void q_wait_regs2(T& a, T& b) {
    if constexpr (N_ == 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b) :: "memory");
    Q_WAIT2_CASE(2); Q_WAIT2_CASE(4); Q_WAIT2_CASE(6); Q_WAIT2_CASE(8); Q_WAIT2_CASE(10); Q_WAIT2_CASE(12);
    Q_WAIT2_CASE(14); Q_WAIT2_CASE(16); Q_WAIT2_CASE(18); Q_WAIT2_CASE(20); Q_WAIT2_CASE(22); Q_WAIT2_CASE(24); Q_WAIT2_CASE(26); Q_WAIT2_CASE(28); Q_WAIT2_CASE(30);
    else static_assert(N_ < 0, "unsupported vmcnt");
}

// 8-wide forms of common.h's gelu_fast2 / gelu_grad_fast2 (same polynomial, same operation order per element): written on
// 8-vectors so that every Horner step is four INDEPENDENT v_pk_fma_f32 -- the 2-wide form compiled to one dependent chain
// per pair with a stall slot after every step.
typedef float q_f32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ q_f32x8 q_splat8(float v) { return q_f32x8{v, v, v, v, v, v, v, v}; }
// Phi of x clamped to [-4.5, 4.5]; also hands back the clamped x and its square (the density term of the derivative reuses them)
__device__ __forceinline__ q_f32x8 q_phi_parts8(q_f32x8 x, q_f32x8& xc, q_f32x8& u) {
#pragma unroll
    for (int e = 0; e < 8; ++e) xc[e] = __builtin_amdgcn_fmed3f(x[e], -4.5f, 4.5f);
    u = xc * xc;
    q_f32x8 q = __builtin_elementwise_fma(u, q_splat8(3.619783835e-11f), q_splat8(-3.842468662e-09f));
    q = __builtin_elementwise_fma(q, u, q_splat8(1.789582063e-07f));
    q = __builtin_elementwise_fma(q, u, q_splat8(-4.853476327e-06f));
    q = __builtin_elementwise_fma(q, u, q_splat8(8.614045158e-05f));
    q = __builtin_elementwise_fma(q, u, q_splat8(-1.069849927e-03f));
    q = __builtin_elementwise_fma(q, u, q_splat8(9.707349039e-03f));
    q = __builtin_elementwise_fma(q, u, q_splat8(-6.620850869e-02f));
    q = __builtin_elementwise_fma(q, u, q_splat8(3.988530737e-01f));
    return __builtin_elementwise_fma(xc, q, q_splat8(0.5f));
}
__device__ __forceinline__ q_f32x8 q_phi8(q_f32x8 x) { q_f32x8 xc, u; return q_phi_parts8(x, xc, u); }
__device__ __forceinline__ q_f32x8 q_gelu8(q_f32x8 x) { return x * q_phi8(x); }
// x phi(x) of the derivative on the CLAMPED x (round 6: the square and the clamp are Phi's own -- 8 VALU instructions per 8 values fewer in
// the VALU-bound FF1 epilogue): identical to the unclamped form for |x| <= 4.5; beyond, |x phi(x)| < 8e-5 either way (the 8-bit code's
// step is 5e-3, a bf16's at 1.0 is 4e-3)
__device__ __forceinline__ q_f32x8 q_gelu_grad8(q_f32x8 x) {
    q_f32x8 xc, u;
    const q_f32x8 ph = q_phi_parts8(x, xc, u);
    const q_f32x8 w = u * q_splat8(-0.72134752044448170368f);
    q_f32x8 e;
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = __builtin_amdgcn_exp2f(w[i]);
    return __builtin_elementwise_fma(xc * q_splat8(0.39894228040143267794f), e, ph);
}
// GELU and its derivative of the same 8 values with ONE evaluation of Phi (the BIAS_GELU_DG epilogue: the forward GEMM stores gelu'(pre) for
// the backward instead of pre, so that the backward's epilogue is a multiplication -- same operations per element as q_gelu8 / q_gelu_grad8)
__device__ __forceinline__ void q_gelu_both8(q_f32x8 x, q_f32x8& g, q_f32x8& dg) {
    q_f32x8 xc, u;
    const q_f32x8 ph = q_phi_parts8(x, xc, u);
    const q_f32x8 w = u * q_splat8(-0.72134752044448170368f);
    q_f32x8 e;
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = __builtin_amdgcn_exp2f(w[i]);
    g = x * ph;
    dg = __builtin_elementwise_fma(xc * q_splat8(0.39894228040143267794f), e, ph);
}
__device__ __forceinline__ q_f32x8 q_widen8(bf16x8 v) {
    q_f32x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (float)v[e];
    return r;
}
__device__ __forceinline__ bf16x8 q_narrow8(q_f32x8 v) {
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (__bf16)v[e];
    return r;
}

template <int EPI> __host__ __device__ constexpr bool q_has_bias() {
    return EPI == VITK_EPI_BIAS || EPI == VITK_EPI_BIAS_GELU || EPI == VITK_EPI_BIAS_GELU_DG || EPI == VITK_EPI_BIAS_GELU_DG8 || EPI == VITK_EPI_RESID ||
           EPI == VITK_EPI_RESID16;
}

// ---- the gelu' factor in 8 bits (VITK_EPI_BIAS_GELU_DG8 stores it, VITK_EPI_MUL_AUX8 multiplies by it) ----------------------------
// FF1's second output and dFF1's second input are 310 MB each at ViT-B/16 batch 256 as 16-bit values, and both epilogues run at the
// memory system's rate.  gelu'(x) lies in [-0.1290, 1.1290]: fixed point  code = rne(200 f) + 27  (1 .. 253),  f~ = 0.005 (code - 27)
// has |f~ - f| <= 0.0025 over the whole range -- a bf16 of the same value is off by up to 0.0039 in [1, 1.13) and 0.0020 in [0.5, 1) --
// and 0, 0.5 and 1 (the tails and x = 0) are exact.
typedef unsigned q_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void q_gload_u32x2(q_u32x2& d, const unsigned char* p) { asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ q_u32x2 q_dg_encode8(q_f32x8 dg) {
    // magic-number rounding: the low mantissa byte of 200 f + 27 + 1.5 * 2^23 is the code (round to nearest even; 0 <= 200 f + 27 <= 255)
    const q_f32x8 t = __builtin_elementwise_fma(dg, q_splat8(200.f), q_splat8(12582939.f));
    unsigned b[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float te = t[e];          // (a __builtin_bit_cast of the vector-element expression itself reads element 0 with this clang)
        b[e] = __builtin_bit_cast(unsigned, te);
    }
    // the low bytes of four words into one word: two byte permutes + one OR (v_perm_b32: selector 0..3 = bytes of the second operand,
    // 4..7 = bytes of the first, 0x0c = 0x00) -- 6 instructions per 8 codes instead of 12 shifts / masks / ors
    auto low4 = [](unsigned w0, unsigned w1, unsigned w2, unsigned w3) {
        return __builtin_amdgcn_perm(w1, w0, 0x0c0c0400u) | __builtin_amdgcn_perm(w3, w2, 0x04000c0cu);
    };
    return q_u32x2{low4(b[0], b[1], b[2], b[3]), low4(b[4], b[5], b[6], b[7])};
}
__device__ __forceinline__ q_f32x8 q_dg_decode8(q_u32x2 c) {
    q_f32x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (float)((c[e >> 2] >> (8 * (e & 3))) & 0xffu);       // v_cvt_f32_ubyteN
    return (r - q_splat8(27.f)) * q_splat8(0.005f);
}
template <int EPI> __host__ __device__ constexpr bool q_aux_in() { return EPI == VITK_EPI_GELU_BWD || EPI == VITK_EPI_MUL_AUX || EPI == VITK_EPI_MUL_AUX8; }
template <int EPI> __host__ __device__ constexpr bool q_two_outputs() { return EPI == VITK_EPI_BIAS_GELU || EPI == VITK_EPI_BIAS_GELU_DG || EPI == VITK_EPI_BIAS_GELU_DG8; }

__device__ __forceinline__ unsigned q_dpp_xor1(unsigned v) {       // value of lane ^ 1 (quad_perm [1,0,3,2])
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);
}
__device__ __forceinline__ unsigned q_pack2(float a, float b) {
    const bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
// two accumulator values (+ their bias) rounded to the 16-bit type and packed: the adds as ONE packed add, and none at all in the epilogues
// without a bias (an `acc + 0.f` is not dropped by the compiler: it turns a -0 into +0)
template <bool HAS_BIAS> __device__ __forceinline__ unsigned q_pack2b(float a, float b, float ba, float bb) {
    if constexpr (HAS_BIAS) {
        const f32x2 t = f32x2{a, b} + f32x2{ba, bb};
        return q_pack2(t[0], t[1]);
    } else return q_pack2(a, b);
}
typedef unsigned q_u32x4 __attribute__((ext_vector_type(4)));
// Column sums of the ROUNDED outputs (what a later colsum(C) would read): cs[e] += (float)g8[e], one v_dot2c_f32 per element -- the pair
// against (1, 0) or (0, 1), accumulated in f32 -- instead of an unpack and an add (round 6).  The products are exact and the other
// lane of the pair contributes 0 x value: a NON-FINITE value in the neighbouring column therefore turns this column's sum into NaN
// too (0 x Inf); a step with a non-finite gradient is lost either way.
__device__ __forceinline__ void q_cs_add8(float (&cs)[8], bf16x8 g8) {
    typedef __bf16 q_h2 __attribute__((ext_vector_type(2)));
    // the unit pairs as OPAQUE register values: hipcc folds a literal {1, 0} pair into the inline constant "1.0", which the instruction
    // reads as the 32-bit pattern 0x3f800000 = the pair (0, 1) [measured: both sums then collect the odd column]
#ifdef VITK_HALF_IS_F16
    unsigned k0 = 0x00003c00u, k1 = 0x3c000000u;
#else
    unsigned k0 = 0x00003f80u, k1 = 0x3f800000u;
#endif
    asm volatile("" : "+s"(k0), "+s"(k1));
    const q_h2 e0 = __builtin_bit_cast(q_h2, k0), e1 = __builtin_bit_cast(q_h2, k1);
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const q_h2 v = {g8[e], g8[e + 1]};
#ifdef VITK_HALF_IS_F16
        cs[e] = __builtin_amdgcn_fdot2(v, e0, cs[e], false);
        cs[e + 1] = __builtin_amdgcn_fdot2(v, e1, cs[e + 1], false);
#else
        cs[e] = __builtin_amdgcn_fdot2_f32_bf16(v, e0, cs[e], false);
        cs[e + 1] = __builtin_amdgcn_fdot2_f32_bf16(v, e1, cs[e + 1], false);
#endif
    }
}

}  // namespace
