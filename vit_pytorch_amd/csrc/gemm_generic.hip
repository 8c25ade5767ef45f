// gemm_generic.hip -- strided batched GEMM with f32 accumulation on the vector ALU.
//
// The coverage kernel: f32 validation mode, extents the MFMA kernels do not take (K % 32 != 0,
// the classifier head, tiny models such as BASELINE config 1), and the materialising attention
// path (QK^T and P.V as batched GEMMs) used when `Attention.attend` carries forward hooks
// (recorder.py:26-29), when dim_head != 64 and in f32 mode.  Not a performance path.
#include "common.h"

namespace {

constexpr int GB = 64, GK = 16;

struct MatD { const void* p; int dt; long long s_b1, s_b2, s_row, s_col; };

__device__ __forceinline__ float ld_any(const void* p, int dt, long long idx) {
    return dt == VITK_F32 ? reinterpret_cast<const float*>(p)[idx] : (float)reinterpret_cast<const __bf16*>(p)[idx];
}
__device__ __forceinline__ void st_any(void* p, int dt, long long idx, float v) {
    if (dt == VITK_F32) reinterpret_cast<float*>(p)[idx] = v;
    else reinterpret_cast<__bf16*>(p)[idx] = (__bf16)v;
}

__global__ __launch_bounds__(256) void gemm_generic_kernel(MatD A, MatD B, MatD C, const void* bias, int bias_dt,
                                                            int nb2, int M, int N, int K, float alpha, float beta) {
    __shared__ float As[GK][GB + 1];
    __shared__ float Bs[GK][GB + 1];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int b = blockIdx.z, b1 = b / nb2, b2 = b % nb2;
    const long long aoff = b1 * A.s_b1 + b2 * A.s_b2;
    const long long boff = b1 * B.s_b1 + b2 * B.s_b2;
    const long long coff = b1 * C.s_b1 + b2 * C.s_b2;
    const int m0 = blockIdx.y * GB, n0 = blockIdx.x * GB;
    const bool a_kfast = A.s_col == 1;   // consecutive threads along k when k is the contiguous index
    const bool b_kfast = B.s_row == 1;

    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    for (int k0 = 0; k0 < K; k0 += GK) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + 256 * e;
            int mm, kk;
            if (a_kfast) { kk = idx & 15; mm = idx >> 4; } else { mm = idx & 63; kk = idx >> 6; }
            const int m = m0 + mm, k = k0 + kk;
            As[kk][mm] = (m < M && k < K) ? ld_any(A.p, A.dt, aoff + m * A.s_row + k * A.s_col) : 0.f;
            int nn, kb;
            if (b_kfast) { kb = idx & 15; nn = idx >> 4; } else { nn = idx & 63; kb = idx >> 6; }
            const int n = n0 + nn, k2 = k0 + kb;
            Bs[kb][nn] = (n < N && k2 < K) ? ld_any(B.p, B.dt, boff + k2 * B.s_row + n * B.s_col) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < GK; ++kk) {
            float a[4], bb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) bb[j] = Bs[kk][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], bb[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n >= N) continue;
            const long long ci = coff + m * C.s_row + n * C.s_col;
            float v = alpha * acc[i][j];
            if (beta != 0.f) v += beta * ld_any(C.p, C.dt, ci);
            if (bias) v += ld_any(bias, bias_dt, n);
            st_any(const_cast<void*>(C.p), C.dt, ci, v);
        }
    }
}

}  // namespace

extern "C" int vitk_gemm_generic(vitk_mat A, vitk_mat B, vitk_mat C, const void* bias, int bias_dt, int64_t nb1,
                                 int64_t nb2, int64_t M, int64_t N, int64_t K, float alpha, float beta, void* stream) {
    if (!A.p || !B.p || !C.p) VITK_FAIL(VITK_E_ARG, "gemm_generic: null pointer");
    if (M <= 0 || N <= 0 || K < 0 || nb1 <= 0 || nb2 <= 0) VITK_FAIL(VITK_E_SHAPE, "gemm_generic: empty problem");
    if (nb1 * nb2 > 65535 || (M + GB - 1) / GB > 65535) VITK_FAIL(VITK_E_SHAPE, "gemm_generic: batch or M too large for the grid");
    for (int dt : {A.dt, B.dt, C.dt}) if (dt != VITK_F32 && dt != VITK_BF16) VITK_FAIL(VITK_E_DTYPE, "gemm_generic: bad dtype");
    if (bias && bias_dt != VITK_F32 && bias_dt != VITK_BF16) VITK_FAIL(VITK_E_DTYPE, "gemm_generic: bad bias dtype");
    const MatD a{A.p, A.dt, A.s_b1, A.s_b2, A.s_row, A.s_col}, b{B.p, B.dt, B.s_b1, B.s_b2, B.s_row, B.s_col},
        c{C.p, C.dt, C.s_b1, C.s_b2, C.s_row, C.s_col};
    const dim3 grid((unsigned)((N + GB - 1) / GB), (unsigned)((M + GB - 1) / GB), (unsigned)(nb1 * nb2));
    hipLaunchKernelGGL(gemm_generic_kernel, grid, dim3(256), 0, (hipStream_t)stream, a, b, c, bias, bias_dt, (int)nb2, (int)M,
                       (int)N, (int)K, alpha, beta);
    VITK_CHECK_LAUNCH("gemm_generic");
    return 0;
}
