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
