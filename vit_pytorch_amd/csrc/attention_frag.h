// attention_frag.h -- fragment helpers shared by the attention kernels (attention.hip, attention_pipe.hip, attention_varlen.hip).
//
// v_mfma_f32_16x16x32_bf16: D[i][j] = sum_k A[i][k] B[k][j]; a lane (fi = lane & 15, fg = lane >> 4) supplies A[fi][8 fg .. + 7],
// B[8 fg .. + 7][fi] and holds D[4 fg + r][fi], r = 0..3.
#pragma once
#include "common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ bf16x8 pack8(f32x4 a, f32x4 b) {
    bf16x8 r = {(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3], (__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
    return r;
}
// An accumulator row set leaves for global memory: lane (fi, fg) holds columns 16 fd + 4 fg .. + 3 of row fi for fd < NFD.  Lane pairs
// (fg, fg ^ 1) exchange halves per pair of 16-column blocks (v_permlane16_swap: odd 16-lane rows of the first operand <-> even rows of the
// second), after which an even-fg lane owns 8 consecutive columns of block 2 j and an odd-fg lane 8 of block 2 j + 1: ONE 16-byte store per
// lane and block pair instead of two 8-byte ones (round 6: beside an LDS-DMA stream it is the number of store instructions that costs;
// profiles/r06ab_store_widening_ab.log).  All 64 lanes must call it (the exchange is not predicated); `ok` masks the stores.  rowp = the
// lane's row (column 0), 16-byte aligned; the values are scaled by `mul` on the way.
template <int NFD> __device__ __forceinline__ void store_rows16(__bf16* rowp, const f32x4 (&v)[NFD], float mul, int fg, bool ok) {
    typedef unsigned fr_u4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int j = 0; j + 1 < NFD; j += 2) {
        const fr_u4 x = __builtin_bit_cast(fr_u4, pack8(v[j] * mul, v[j + 1] * mul));       // dwords 0, 1: block j; 2, 3: block j + 1
        auto s0 = __builtin_amdgcn_permlane16_swap(x[0], x[2], false, false);
        auto s1 = __builtin_amdgcn_permlane16_swap(x[1], x[3], false, false);
        if (ok) *reinterpret_cast<fr_u4*>(rowp + 16 * j + 16 * (fg & 1) + 8 * (fg >> 1)) = fr_u4{(unsigned)s0[0], (unsigned)s1[0], (unsigned)s0[1], (unsigned)s1[1]};
    }
    if constexpr (NFD & 1) {
        if (ok) {
            const f32x4 t = v[NFD - 1] * mul;
            const bf16x4 o4 = {(__bf16)t[0], (__bf16)t[1], (__bf16)t[2], (__bf16)t[3]};
            *reinterpret_cast<bf16x4*>(rowp + 16 * (NFD - 1) + 4 * fg) = o4;
        }
    }
}
__device__ __forceinline__ float dot8(bf16x8 a, bf16x8 b) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += (float)a[e] * (float)b[e];
    return s;
}
// Reductions over the four 16-lane groups of a wave (lanes l, l^16, l^32, l^48 hold the same query/key column):
// v_permlane16_swap / v_permlane32_swap exchange the groups in the VALU -- no LDS round trip and no
// s_waitcnt lgkmcnt(0) in the middle of the fragment reads, unlike ds_bpermute (__shfl_xor).
__device__ __forceinline__ float groups_max(float x) {
    unsigned u = __builtin_bit_cast(unsigned, x);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    x = fmaxf(__builtin_bit_cast(float, (unsigned)a[0]), __builtin_bit_cast(float, (unsigned)a[1]));
    u = __builtin_bit_cast(unsigned, x);
    auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)b[0]), __builtin_bit_cast(float, (unsigned)b[1]));
}
__device__ __forceinline__ float groups_sum(float x) {
    unsigned u = __builtin_bit_cast(unsigned, x);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    x = __builtin_bit_cast(float, (unsigned)a[0]) + __builtin_bit_cast(float, (unsigned)a[1]);
    u = __builtin_bit_cast(unsigned, x);
    auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)b[0]) + __builtin_bit_cast(float, (unsigned)b[1]);
}

}  // namespace
