// layernorm.hip -- LayerNorm forward/backward and column-sum reductions for gfx950.
//
// Replaces nn.LayerNorm (vit.py:19,39,69,101,103; simple_vit.py:29,42,67,92,94) and its autograd.
// HBM-bound: one wave64 per row, the row lives in registers (4 elements per lane per chunk,
// 16-byte loads for f32 / 8-byte for bf16), statistics by wave shuffles, f32 arithmetic.
// Algorithmic bytes per row: D*(sizeof(x)+sizeof(y)) forward; backward reads dy and x and
// writes dx (f32 and/or T).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int LN_THREADS = 256;
constexpr int LN_WAVES = LN_THREADS / WAVE;
// backward: 8 waves per block, up to 512 blocks = 16 waves per CU (4 per SIMD at <= 128 VGPRs)
constexpr int LNB_THREADS = 512;
constexpr int LNB_WAVES = LNB_THREADS / WAVE;
constexpr int LNB_MAX_BLOCKS = 512;   // two 8-wave blocks per CU: all resident at once; 1024 measured the same kernel time and doubles the partial rows the finalize reads

// streamed-once row loads (the LayerNorm backward gained 123 -> 99 us from the same hint)
template <typename T> __device__ __forceinline__ f32x4 load4_nt(const T* p);
template <> __device__ __forceinline__ f32x4 load4_nt<float>(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); }
template <> __device__ __forceinline__ f32x4 load4_nt<__bf16>(const __bf16* p) {
    const bf16x4 v = __builtin_nontemporal_load(reinterpret_cast<const bf16x4*>(p));
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
template <typename XT, typename YT, typename WT, int MAXC, bool NT>
__global__ __launch_bounds__(LN_THREADS) void ln_fwd_kernel(
    const XT* __restrict__ x, const WT* __restrict__ w, const WT* __restrict__ b,
    YT* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out,
    long long rows, int D, float eps, RowMap imap, RowMap omap,
    const WT* __restrict__ add, long long add_group, long long add_off, F8Out f8) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nchunk = D >> 2;
    const float invD = 1.0f / (float)D;
    const float sc8 = f8.p ? *f8.scale : 0.f;
    float amax = 0.f;

    f32x4 wv[MAXC], bv[MAXC];
#pragma unroll
    for (int t = 0; t < MAXC; ++t) {
        const int c = lane + 64 * t;
        if (c < nchunk) {
            wv[t] = load4<WT>(w + 4 * c);
            bv[t] = b ? load4<WT>(b + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // The NEXT row of this wave is requested before the current one is reduced and stored: two rows per wave in flight (a wave with
    // one row spends the row's load latency idle; [measured, 50,432 x 768 f32 -> bf16] 48.6 -> 36.8 us = 6.3 TB/s)
    const long long stride = (long long)gridDim.x * LN_WAVES;
    auto load_row = [&](long long row, f32x4 (&dst)[MAXC]) {
        const XT* xr = x + map_row(imap, row) * (long long)D;
#pragma unroll
        for (int t = 0; t < MAXC; ++t) {
            const int c = lane + 64 * t;
            if (c < nchunk) { if constexpr (NT) dst[t] = load4_nt<XT>(xr + 4 * c); else dst[t] = load4<XT>(xr + 4 * c); }
        }
    };
    f32x4 vn[MAXC];
    long long row = (long long)blockIdx.x * LN_WAVES + wave;
    if (row < rows) load_row(row, vn);
    for (; row < rows; row += stride) {
        f32x4 v[MAXC];
#pragma unroll
        for (int t = 0; t < MAXC; ++t) v[t] = vn[t];
        if (row + stride < rows) load_row(row + stride, vn);
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < MAXC; ++t) {
            const int c = lane + 64 * t;
            if (c < nchunk) s += (v[t][0] + v[t][1]) + (v[t][2] + v[t][3]);
        }
        const float mean = wave_sum(s) * invD;
        float q = 0.f;
#pragma unroll
        for (int t = 0; t < MAXC; ++t) {
            const int c = lane + 64 * t;
            if (c < nchunk) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[t][e] - mean; q += d * d; }
            }
        }
        const float var = wave_sum(q) * invD;
        const float rstd = 1.0f / sqrtf(var + eps);
        YT* yr = y + map_row(omap, row) * (long long)D;
        const WT* ar = add ? add + ((add_group > 0 ? row % add_group : row) + add_off) * (long long)D : nullptr;
#pragma unroll
        for (int t = 0; t < MAXC; ++t) {
            const int c = lane + 64 * t;
            if (c < nchunk) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (v[t][e] - mean) * rstd * wv[t][e] + bv[t][e];
                if (ar) { const f32x4 a = load4<WT>(ar + 4 * c); o += a; }
                store4<YT>(yr + 4 * c, o);
                if (f8.p) reinterpret_cast<unsigned*>(f8.p + map_row(omap, row) * (long long)D)[c] = pack_fp8x4(o, sc8);
                if (f8.amax) amax = fmaxf(amax, absmax4(o));
            }
        }
        if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
    }
    if (f8.amax) {     // one atomic per BLOCK (the launcher also caps the grid at 1024 blocks): same-address atomics retire at
                       // ~4 ns each in L2 -- one per wave of an 8192-block grid quadrupled the kernel's duration
        __shared__ float red[LN_WAVES];
        amax = wave_max(amax);
        if (lane == 0) red[wave] = amax;
        __syncthreads();
        if (threadIdx.x == 0) {
            float m = red[0];
#pragma unroll
            for (int k = 1; k < LN_WAVES; ++k) m = fmaxf(m, red[k]);
            atomicMax(f8.amax + (blockIdx.x & 63), __builtin_bit_cast(unsigned, m));
        }
    }
}

// ---- 16-bit rows, TWO ROWS PER WAVE (round 4: the forward residual stream is 16-bit by default, ops.fwd_stream_16) -------------------
// ln_fwd_kernel above moves a 16-bit row as 8-byte pieces (4 elements per lane and chunk): with the stream in the parameter dtype its
// bytes fell from 232 to 155 MB at ViT-B/16 but its time did not (37 -> 38.7 us in the step, 4.0 TB/s: 8-byte accesses run at
// 0.54-0.70x the 16-byte rate, MI355X_MICROARCH.md).  Here a row of D = 256 * CPL elements is owned by HALF a wave: lane l of the half reads
// CPL chunks of 8 elements (16 bytes) at chunk l + 32 t, so every load / store instruction moves 2 x 512 contiguous bytes; the statistics
// are reduced inside the half (DPP rotations of the 16-lane rows + one v_permlane16_swap -- no v_permlane32_swap); the NEXT pair of rows
// is requested before the current pair is reduced (four rows per wave in flight).  Serves x, y, w (b) of one 16-bit type, identity row
// maps, no fused add, no fp8 side output -- the transformer layers' LayerNorms; everything else stays on ln_fwd_kernel.
__device__ __forceinline__ float half_wave_sum(float v) {
    v += dpp_f32<0x128>(v);   // row_ror:8
    v += dpp_f32<0x124>(v);   // row_ror:4
    v += dpp_f32<0x122>(v);   // row_ror:2
    v += dpp_f32<0x121>(v);   // row_ror:1
    const unsigned u = __builtin_bit_cast(unsigned, v);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);      // rows 0 <-> 1 and 2 <-> 3: each half of the wave sums its two rows
    return __builtin_bit_cast(float, (unsigned)a[0]) + __builtin_bit_cast(float, (unsigned)a[1]);
}
typedef float ln_f32x8 __attribute__((ext_vector_type(8)));
template <int CPL, bool HASB>
__global__ __launch_bounds__(LN_THREADS) void ln_fwd16_kernel(
    const __bf16* __restrict__ x, const __bf16* __restrict__ w, const __bf16* __restrict__ b, __bf16* __restrict__ y,
    float* __restrict__ mean_out, float* __restrict__ rstd_out, long long rows, float eps) {
    constexpr int D = 256 * CPL;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hl = lane & 31, half = lane >> 5;
    const float invD = 1.0f / (float)D;
    ln_f32x8 wv[CPL], bv[CPL];
#pragma unroll
    for (int t = 0; t < CPL; ++t) {
        const bf16x8 w8 = *reinterpret_cast<const bf16x8*>(w + 8 * (hl + 32 * t));
#pragma unroll
        for (int e = 0; e < 8; ++e) wv[t][e] = (float)w8[e];
        if constexpr (HASB) {
            const bf16x8 b8 = *reinterpret_cast<const bf16x8*>(b + 8 * (hl + 32 * t));
#pragma unroll
            for (int e = 0; e < 8; ++e) bv[t][e] = (float)b8[e];
        }
    }
    const long long stride = (long long)gridDim.x * LN_WAVES * 2;
    long long row = ((long long)blockIdx.x * LN_WAVES + wave) * 2 + half;
    auto load_row = [&](long long r, bf16x8 (&dst)[CPL]) {
        const __bf16* xr = x + (r < rows ? r : rows - 1) * (long long)D + 8 * hl;       // a half past the end re-reads the last row, stores nothing
#pragma unroll
        for (int t = 0; t < CPL; ++t) dst[t] = *reinterpret_cast<const bf16x8*>(xr + 256 * t);
    };
    bf16x8 nx[CPL];
    load_row(row, nx);
    // the loop bound is wave-uniform (the even row of the pair): both halves run the same number of iterations
    for (long long r0 = row - half; r0 < rows; r0 += stride, row += stride) {
        ln_f32x8 v[CPL];
#pragma unroll
        for (int t = 0; t < CPL; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[t][e] = (float)nx[t][e];
        if (r0 + stride < rows) load_row(row + stride, nx);
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < CPL; ++t) s += ((v[t][0] + v[t][1]) + (v[t][2] + v[t][3])) + ((v[t][4] + v[t][5]) + (v[t][6] + v[t][7]));
        const float mean = half_wave_sum(s) * invD;
        float q = 0.f;
#pragma unroll
        for (int t = 0; t < CPL; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[t][e] - mean; q += d * d; }
        const float var = half_wave_sum(q) * invD;
        const float rstd = 1.0f / sqrtf(var + eps);
        if (row < rows) {
            __bf16* yr = y + row * (long long)D + 8 * hl;
#pragma unroll
            for (int t = 0; t < CPL; ++t) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float f = (v[t][e] - mean) * rstd * wv[t][e];
                    if constexpr (HASB) f += bv[t][e];
                    o[e] = (__bf16)f;
                }
                *reinterpret_cast<bf16x8*>(yr + 256 * t) = o;
            }
            if (hl == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
        }
    }
}

// ---- patch embedding, first stage: Rearrange 'b c (h p1) (w p2) -> b (h w) (p1 p2 c)' + LayerNorm(patch_dim) (vit.py:100-101) with the
// GATHER IN THE LOAD -- no `patches` tensor, no patchify pass (north star: "patch-embedding (im2col-into-GEMM)").  16-bit images of 3
// channels, 16 x 16 patches (patch_dim 768 = 32 lanes x 24 elements): half a wave owns a patch; lane l of the half owns pixel row
// i = l >> 1 and the 8 pixels j0 = 8 (l & 1) .. + 7 of it, i.e. three 16-byte loads (one per channel plane), and these 24 values
// are exactly the 24 CONSECUTIVE elements (i * 16 + j0) * 3 .. + 23 of the patch vector ("for each pixel the 3 channels are
// adjacent") -- the channel interleave is a renaming of registers in the f32 arithmetic, and the normalised row is written as
// 32 lanes x 48 contiguous bytes.  Two horizontally adjacent patches per wave: 64 contiguous bytes per (channel, pixel row).
template <bool HASB>
__global__ __launch_bounds__(LN_THREADS) void patch_ln_fwd16_kernel(
    const __bf16* __restrict__ img, const __bf16* __restrict__ w, const __bf16* __restrict__ b, __bf16* __restrict__ y,
    float* __restrict__ mean_out, float* __restrict__ rstd_out, long long rows, int hp, int wp, long long plane, int W, float eps) {
    constexpr int D = 768;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hl = lane & 31, half = lane >> 5;
    const int pi = hl >> 1, pj0 = (hl & 1) * 8;
    const float invD = 1.0f / (float)D;
    float wv[24], bv[24];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const bf16x8 w8 = *reinterpret_cast<const bf16x8*>(w + 24 * hl + 8 * t);
#pragma unroll
        for (int e = 0; e < 8; ++e) wv[8 * t + e] = (float)w8[e];
        if constexpr (HASB) {
            const bf16x8 b8 = *reinterpret_cast<const bf16x8*>(b + 24 * hl + 8 * t);
#pragma unroll
            for (int e = 0; e < 8; ++e) bv[8 * t + e] = (float)b8[e];
        }
    }
    const long long stride = (long long)gridDim.x * LN_WAVES * 2;
    long long row = ((long long)blockIdx.x * LN_WAVES + wave) * 2 + half;
    const int per_img = hp * wp;
    auto load_patch = [&](long long r, bf16x8 (&dst)[3]) {
        const long long rr = r < rows ? r : rows - 1;
        const long long bi = rr / per_img;
        const int hw = (int)(rr - bi * per_img);
        const int ph = hw / wp, pw = hw - ph * wp;
        const __bf16* src = img + bi * 3 * plane + (long long)(ph * 16 + pi) * W + pw * 16 + pj0;
#pragma unroll
        for (int c = 0; c < 3; ++c) dst[c] = *reinterpret_cast<const bf16x8*>(src + c * plane);
    };
    bf16x8 nx[3];
    load_patch(row, nx);
    for (long long r0 = row - half; r0 < rows; r0 += stride, row += stride) {
        float v[24];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj)
#pragma unroll
            for (int c = 0; c < 3; ++c) v[3 * jj + c] = (float)nx[c][jj];
        if (r0 + stride < rows) load_patch(row + stride, nx);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 24; ++e) s += v[e];
        const float mean = half_wave_sum(s) * invD;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 24; ++e) { const float d = v[e] - mean; q += d * d; }
        const float rstd = 1.0f / sqrtf(half_wave_sum(q) * invD + eps);
        if (row < rows) {
            __bf16* yr = y + row * (long long)D + 24 * hl;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float f = (v[8 * t + e] - mean) * rstd * wv[8 * t + e];
                    if constexpr (HASB) f += bv[8 * t + e];
                    o[e] = (__bf16)f;
                }
                *reinterpret_cast<bf16x8*>(yr + 8 * t) = o;
            }
            if (hl == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
        }
    }
}

// Its backward: only the parameter gradients exist (the image needs none): dgamma[e] = sum_rows dy[r][e] * xhat[r][e], dbeta[e] = sum_rows
// dy[r][e], with xhat re-formed from the image by the same gather.  Per-lane accumulators for its 24 columns, the 8 half-waves of a block
// folded through LDS, one partial row pair per block (fold with vitk_colsum_partials: partials = [2][nblk][768] floats).
__global__ __launch_bounds__(LN_THREADS) void patch_ln_bwd16_kernel(
    const __bf16* __restrict__ dy, const __bf16* __restrict__ img, const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
    float* __restrict__ partials, long long rows, int hp, int wp, long long plane, int W) {
    constexpr int D = 768;
    __shared__ float red[8 * D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hl = lane & 31, half = lane >> 5;
    const int pi = hl >> 1, pj0 = (hl & 1) * 8;
    const int per_img = hp * wp;
    float aw[24], ab[24];
#pragma unroll
    for (int e = 0; e < 24; ++e) { aw[e] = 0.f; ab[e] = 0.f; }
    const long long stride = (long long)gridDim.x * LN_WAVES * 2;
    for (long long row = ((long long)blockIdx.x * LN_WAVES + wave) * 2 + half; row < rows; row += stride) {
        const long long bi = row / per_img;
        const int hw = (int)(row - bi * per_img);
        const int ph = hw / wp, pw = hw - ph * wp;
        const __bf16* src = img + bi * 3 * plane + (long long)(ph * 16 + pi) * W + pw * 16 + pj0;
        bf16x8 xv[3], gv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) xv[c] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(src + c * plane));
        const __bf16* gr = dy + row * (long long)D + 24 * hl;
#pragma unroll
        for (int t = 0; t < 3; ++t) gv[t] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(gr + 8 * t));
        const float mean = mean_in[row], rstd = rstd_in[row];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int e = 3 * jj + c;
                const float g = (float)gv[e >> 3][e & 7];
                aw[e] += g * (((float)xv[c][jj] - mean) * rstd);
                ab[e] += g;
            }
    }
    float* mine = red + (wave * 2 + half) * D + 24 * hl;
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int e = 0; e < 24; ++e) mine[e] = pass ? ab[e] : aw[e];
        __syncthreads();
        for (int c = threadIdx.x; c < D; c += LN_THREADS) {
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) sum += red[k * D + c];
            partials[((long long)pass * gridDim.x + blockIdx.x) * D + c] = sum;
        }
        __syncthreads();
    }
}

// Backward.  DXT is the dtype of the optional second dx output (dx_t).
// NW waves per block: 8 for rows up to 768 columns (<= 128 VGPRs: two blocks per CU), 4 for wider rows, whose three
// column accumulators push the kernel to ~155-170 VGPRs -- three 4-wave blocks then fit a CU (12 waves) where a single
// 8-wave block would (8 waves).
// GT: type of the incoming stream gradient `gin` -- float (f32 stream: dx_f32 is its update, dx_t the 16-bit copy the GEMMs read) or
// DXT (16-bit stream: dx_t IS the updated stream, dx_f32 is null).
template <typename DYT, typename XT, typename WT, typename DXT, typename GT, int MAXC, int NW>
__global__ __launch_bounds__(NW * WAVE, (MAXC <= 3 ? 4 : 1)) void ln_bwd_kernel(
    const DYT* __restrict__ dy, const XT* __restrict__ x, const WT* __restrict__ w,
    const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
    const GT* __restrict__ gin, float* __restrict__ dx_f32, DXT* __restrict__ dx_t,
    float* __restrict__ partials, int colsum_dx,
    long long rows, int D, RowMap dymap, RowMap xmap, RowMap dxmap, unsigned drop_t, unsigned drop_seed, float inv_keep) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nchunk = D >> 2;
    const float invD = 1.0f / (float)D;
    const bool want_dx = (dx_f32 != nullptr) || (dx_t != nullptr);

    f32x4 wv[MAXC], acc_w[MAXC], acc_b[MAXC], acc_x[MAXC];
#pragma unroll
    for (int t = 0; t < MAXC; ++t) {
        const int c = lane + 64 * t;
        if (c < nchunk) wv[t] = load4<WT>(w + 4 * c);
        acc_w[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc_b[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc_x[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (long long row = (long long)blockIdx.x * NW + wave; row < rows; row += (long long)gridDim.x * NW) {
        const DYT* dyr = dy + map_row(dymap, row) * (long long)D;
        const XT* xr = x + map_row(xmap, row) * (long long)D;
        const float mean = mean_in[row], rstd = rstd_in[row];
        const long long orow = map_row(dxmap, row);
        f32x4 g[MAXC], xh[MAXC], gi[MAXC];
        float s1 = 0.f, s2 = 0.f;
        // every load of the row -- incoming stream gradient included -- is issued before the first reduction, so a row
        // costs one memory round trip, not two
        constexpr bool EARLY_GIN = MAXC <= 3;     // wider rows: the 4*MAXC extra registers would cost a wave per SIMD
#pragma unroll
        for (int t = 0; t < MAXC; ++t) {
            const int c = lane + 64 * t;
            gi[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (EARLY_GIN && gin && want_dx && c < nchunk) gi[t] = load4<GT>(gin + orow * D + 4 * c);
        }
#pragma unroll
        for (int t = 0; t < MAXC; ++t) {
            const int c = lane + 64 * t;
            if (c < nchunk) {
                const f32x4 d = load4<DYT>(dyr + 4 * c);
                const f32x4 xv = load4<XT>(xr + 4 * c);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[t][e] = (xv[e] - mean) * rstd;
                    g[t][e] = d[e] * wv[t][e];
                    s1 += g[t][e];
                    s2 += g[t][e] * xh[t][e];
                    acc_w[t][e] += d[e] * xh[t][e];
                    acc_b[t][e] += d[e];
                }
            }
        }
        if (want_dx) {
            const float c1 = wave_sum(s1) * invD;
            const float c2 = wave_sum(s2) * invD;
#pragma unroll
            for (int t = 0; t < MAXC; ++t) {
                const int c = lane + 64 * t;
                if (c < nchunk) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = rstd * (g[t][e] - c1 - xh[t][e] * c2);
                    if (EARLY_GIN) o += gi[t];
                    else if (gin) o += load4<GT>(gin + orow * D + 4 * c);
                    if (dx_f32) *reinterpret_cast<f32x4*>(dx_f32 + orow * D + 4 * c) = o;
                    if (drop_t) {           // dx_t / its column sums are the gradient at the OUTPUT of the preceding Linear, whose
                                            // dropout (vit.py:24,48) kept element (row, col) by the same hash; the f32 stream is not masked
                        const unsigned hrow = drop_row((unsigned)orow, drop_seed);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = drop_keep(hrow, (unsigned)(4 * c + e), drop_t) ? o[e] * inv_keep : 0.f;
                    }
                    if (dx_t) store4<DXT>(dx_t + orow * D + 4 * c, o);
                    if (colsum_dx) acc_x[t] += o;
                }
            }
        }
    }
    // block reduce the per-wave column accumulators through LDS, one slab at a time
    __shared__ f32x4 red[NW][64];
    const int nslab = colsum_dx ? 3 : 2;
    for (int slab = 0; slab < nslab; ++slab) {
#pragma unroll
        for (int t = 0; t < MAXC; ++t) {
            if (64 * t < nchunk) {  // uniform across the block
                const f32x4 a = slab == 0 ? acc_w[t] : (slab == 1 ? acc_b[t] : acc_x[t]);
                red[wave][lane] = a;
                __syncthreads();
                if (wave == 0) {
                    const int c = lane + 64 * t;
                    if (c < nchunk) {
                        f32x4 sum = red[0][lane];
#pragma unroll
                        for (int wv_ = 1; wv_ < NW; ++wv_) sum += red[wv_][lane];
                        *reinterpret_cast<f32x4*>(partials + ((long long)slab * gridDim.x + blockIdx.x) * D + 4 * c) = sum;
                    }
                }
                __syncthreads();
            }
        }
    }
}

template <bool NT> __device__ __forceinline__ f32x4 ld16(const float* p) {
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return *reinterpret_cast<const f32x4*>(p);
}

// The transformer-layer case of ln_bwd_kernel, specialised: full chunks (D == 256 * MAXC), identity row maps, stream gradient
// in, both dx outputs and the column sums of dx_t out, no dropout.  The general kernel spends most of its issue slots on
// predicates it does not need here (a branch around every chunk, five optional-pointer tests per chunk, the row-map
// divisions); this one keeps the row index in SGPRs (readfirstlane), so the per-row statistics are scalar loads and every
// vector access is base + lane offset + immediate, and does its arithmetic on float pairs (v_pk_fma_f32 / v_pk_mul_f32).
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <typename T> struct Raw4;
template <> struct Raw4<float> { using type = f32x4; };
template <> struct Raw4<__bf16> { using type = bf16x4; };
template <bool NT, typename T> __device__ __forceinline__ typename Raw4<T>::type ldraw(const T* p) {
    using R = typename Raw4<T>::type;
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const R*>(p));
    return *reinterpret_cast<const R*>(p);
}
// keeps a loop-invariant packed value packed: without it the compiler hoists the unpacked copy out of the loop (and spills)
__device__ __forceinline__ void opaque(bf16x4& v) {
    unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    asm volatile("" : "+v"(u));
    v = __builtin_bit_cast(bf16x4, u);
}
__device__ __forceinline__ void opaque(f32x4&) {}
__device__ __forceinline__ f32x2 widen2(const f32x4& v, int p) { return f32x2{v[2 * p], v[2 * p + 1]}; }
__device__ __forceinline__ f32x2 widen2(const bf16x4& v, int p) { return f32x2{(float)v[2 * p], (float)v[2 * p + 1]}; }
template <typename DYT, typename XT, typename WT, typename DXT, typename GT, int MAXC, int NW, int WPE, bool NT, bool PIPE, bool DROP>
__global__ __launch_bounds__(NW * WAVE, WPE) void ln_bwd_fast_kernel(
    const DYT* __restrict__ dy, const XT* __restrict__ x, const WT* __restrict__ w,
    const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
    const GT* __restrict__ gin, float* __restrict__ dx_f32, DXT* __restrict__ dx_t,
    float* __restrict__ partials, long long rows, unsigned drop_t, unsigned drop_seed, float inv_keep) {
    constexpr int D = 256 * MAXC;
    constexpr float invD = 1.0f / (float)D;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    typename Raw4<WT>::type wraw[MAXC];          // gamma stays in its storage type: half the registers, one unpack per use
    f32x2 aw[MAXC][2], ab[MAXC][2], ax[MAXC][2];
#pragma unroll
    for (int t = 0; t < MAXC; ++t) {
        wraw[t] = ldraw<false>(w + 4 * (lane + 64 * t));
#pragma unroll
        for (int p = 0; p < 2; ++p) { aw[t][p] = f32x2{0.f, 0.f}; ab[t][p] = f32x2{0.f, 0.f}; ax[t][p] = f32x2{0.f, 0.f}; }
    }
    const long long rstep = (long long)gridDim.x * NW;
    long long row = (long long)blockIdx.x * NW + wave;
    // software pipeline: the loads of row i+1 are issued as soon as row i has left the registers they land in (dy and x after
    // the first pass, the stream gradient after the second), so they fly under row i's reductions, second pass and stores
    constexpr bool F32_STREAM = std::is_same<GT, float>::value;       // else: 16-bit stream, dx_t is its update, no f32 output
    typename Raw4<GT>::type gi[MAXC];
    typename Raw4<DYT>::type dv[MAXC];
    typename Raw4<XT>::type xv[MAXC];
    float mean = 0.f, rstd = 0.f;
    if (PIPE && row < rows) {
        const long long ro = row * D + 4 * lane;
#pragma unroll
        for (int t = 0; t < MAXC; ++t) gi[t] = ldraw<NT>(gin + ro + 256 * t);
#pragma unroll
        for (int t = 0; t < MAXC; ++t) { dv[t] = ldraw<NT>(dy + ro + 256 * t); xv[t] = ldraw<NT>(x + ro + 256 * t); }
        mean = mean_in[row]; rstd = rstd_in[row];
    }
    for (; row < rows; row += rstep) {
        const long long ro = row * D + 4 * lane;
        // the last iteration prefetches its own row again (in bounds, never used): no branch around the loads
        const long long nrow = row + rstep < rows ? row + rstep : row;    // scalar
        const long long nro = nrow * D + 4 * lane;
        if (!PIPE) {
#pragma unroll
            for (int t = 0; t < MAXC; ++t) gi[t] = ldraw<NT>(gin + ro + 256 * t);
#pragma unroll
            for (int t = 0; t < MAXC; ++t) { dv[t] = ldraw<NT>(dy + ro + 256 * t); xv[t] = ldraw<NT>(x + ro + 256 * t); }
            mean = mean_in[row]; rstd = rstd_in[row];
        }
        const float nm = -mean * rstd;
        const f32x2 rs2 = {rstd, rstd}, nm2 = {nm, nm};
        const float rstd_c = rstd;
        f32x2 g[MAXC][2], h[MAXC][2];
        f32x2 s1v = {0.f, 0.f}, s2v = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < MAXC; ++t) {
            opaque(wraw[t]);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const f32x2 d = widen2(dv[t], p);
                const f32x2 xx = widen2(xv[t], p);
                h[t][p] = xx * rs2 + nm2;
                g[t][p] = d * widen2(wraw[t], p);
                s1v += g[t][p];
                s2v += g[t][p] * h[t][p];
                aw[t][p] += d * h[t][p];
                ab[t][p] += d;
            }
        }
        if (PIPE) {
#pragma unroll
            for (int t = 0; t < MAXC; ++t) { dv[t] = ldraw<NT>(dy + nro + 256 * t); xv[t] = ldraw<NT>(x + nro + 256 * t); }
            mean = mean_in[nrow]; rstd = rstd_in[nrow];
        }
        const float c1r = wave_sum(s1v[0] + s1v[1]) * (invD * rstd_c);
        const float c2r = wave_sum(s2v[0] + s2v[1]) * (invD * rstd_c);
        const f32x2 c1 = {c1r, c1r}, c2 = {c2r, c2r};
        float* of = dx_f32 + ro;
        DXT* ot = dx_t + ro;
#pragma unroll
        for (int t = 0; t < MAXC; ++t) {
            f32x2 o[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                o[p] = g[t][p] * rs2 - c1;
                o[p] = o[p] - h[t][p] * c2;
                o[p] += widen2(gi[t], p);
                if (!DROP) ax[t][p] += o[p];
            }
            const f32x4 ov = {o[0][0], o[0][1], o[1][0], o[1][1]};
            if (PIPE) gi[t] = ldraw<NT>(gin + nro + 256 * t);
            if constexpr (F32_STREAM) *reinterpret_cast<f32x4*>(of + 256 * t) = ov;
            if (DROP) {     // dx_t and its column sums are the gradient at the OUTPUT of the preceding Linear, whose dropout
                            // (vit.py:24,48) kept element (row, col) by the same hash; the f32 stream is not masked
                const unsigned hrow = drop_row((unsigned)row, drop_seed);
                f32x4 om;
#pragma unroll
                for (int e = 0; e < 4; ++e) om[e] = drop_keep(hrow, (unsigned)(4 * (lane + 64 * t) + e), drop_t) ? ov[e] * inv_keep : 0.f;
                ax[t][0] += f32x2{om[0], om[1]}; ax[t][1] += f32x2{om[2], om[3]};
                store4<DXT>(ot + 256 * t, om);
            } else store4<DXT>(ot + 256 * t, ov);
        }
    }
    // block reduce the per-wave column accumulators through LDS, one slab at a time (same partial layout as ln_bwd_kernel)
    __shared__ f32x4 red[NW][64];
    for (int slab = 0; slab < 3; ++slab) {
#pragma unroll
        for (int t = 0; t < MAXC; ++t) {
            const f32x2 a0 = slab == 0 ? aw[t][0] : (slab == 1 ? ab[t][0] : ax[t][0]);
            const f32x2 a1 = slab == 0 ? aw[t][1] : (slab == 1 ? ab[t][1] : ax[t][1]);
            red[wave][lane] = f32x4{a0[0], a0[1], a1[0], a1[1]};
            __syncthreads();
            if (wave == 0) {
                f32x4 sum = red[0][lane];
#pragma unroll
                for (int wv_ = 1; wv_ < NW; ++wv_) sum += red[wv_][lane];
                *reinterpret_cast<f32x4*>(partials + ((long long)slab * gridDim.x + blockIdx.x) * D + 4 * (lane + 64 * t)) = sum;
            }
            __syncthreads();
        }
    }
}

// One thread's walk over its share of the partial rows (p = ph, ph + 16, ...), the sums taken in row order as before but with EIGHT loads
// in flight (round 6): the plain `s += src[...]` loop waited for every load before it issued the next -- 25 dependent round trips for the
// 394 partial rows of ViT-B's FF1 bias gradient = the 18 us a fold launch took; same additions in the same order, bit-identical results.
__device__ __forceinline__ float fold_walk(const float* __restrict__ src, long long ld, int nparts, int ph, long long c) {
    float s = 0.f;
    int p = ph;
    for (; p + 7 * 16 < nparts; p += 8 * 16) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = src[(long long)(p + 16 * k) * ld + c];
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[k];
    }
    for (; p < nparts; p += 16) s += src[(long long)p * ld + c];
    return s;
}

// out[c] = (acc ? out[c] : 0) + sum_p partials[p*ld + c]
template <typename OT>
__global__ __launch_bounds__(1024) void colsum_partials_kernel(const float* __restrict__ partials, long long nparts,
                                                                long long ld, long long cols, OT* __restrict__ out, int accumulate) {
    // 64 columns x 16 partial phases per block: the lists are a few hundred rows long (394 for the FF1 bias gradient
    // of ViT-B), so the walk over them is what takes the time -- 16 rows in flight per column instead of 4
    __shared__ float red[16][64];
    const int cx = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const long long c = (long long)blockIdx.x * 64 + cx;
    const float s = c < cols ? fold_walk(partials, ld, (int)nparts, ph, c) : 0.f;
    red[ph][cx] = s;
    __syncthreads();
    if (ph == 0 && c < cols) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][cx];
        if (accumulate) t += to_f32<OT>(out[c]);
        out[c] = from_f32<OT>(t);
    }
}

// Many folds in one launch (round 5): the backward of a layer ends in ~4 tiny fold launches -- two LayerNorm finalizes (3 slabs each), the
// FeedForward bias-gradient column sums -- each 4-11 us of launch latency around a few MB of reads; their outputs are parameter gradients
// nobody reads before the optimizer / the all-reduce, so the engine queues them and folds a layer's worth in ONE launch.  Job j:
// dst[c] = (accumulate ? dst[c] : 0) + sum_p src[p * ld + c], c < cols, p < nparts -- the SAME order of additions as the kernels above
// (16 phases strided over the partial rows, then the 16 phase sums in order), so results are bit-identical to the one-job launches.
constexpr int FM_MAX = 40;
struct FoldMany { const float* src[FM_MAX]; void* dst[FM_MAX]; int nparts[FM_MAX]; int ld[FM_MAX]; int cols[FM_MAX]; int flags[FM_MAX]; int blk0[FM_MAX + 1]; int count; };
__global__ __launch_bounds__(1024) void fold_many_kernel(const FoldMany a) {
    __shared__ float red[16][64];
    int lo = 0, hi = a.count;                  // uniform binary search: blk0[lo] <= blockIdx.x < blk0[lo + 1]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int)blockIdx.x >= a.blk0[mid]) lo = mid; else hi = mid; }
    const float* __restrict__ src = a.src[lo];
    const int nparts = a.nparts[lo], cols = a.cols[lo], flags = a.flags[lo];
    const long long ld = a.ld[lo];
    const int cx = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c = ((int)blockIdx.x - a.blk0[lo]) * 64 + cx;
    const float s = c < cols ? fold_walk(src, ld, nparts, ph, c) : 0.f;
    red[ph][cx] = s;
    __syncthreads();
    if (ph == 0 && c < cols) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][cx];
        if ((flags >> 4) == VITK_F32) {
            float* o = (float*)a.dst[lo];
            if (flags & 1) t += o[c];
            o[c] = t;
        } else {
            __bf16* o = (__bf16*)a.dst[lo];
            if (flags & 1) t += (float)o[c];
            o[c] = (__bf16)t;
        }
    }
}

// LayerNorm-backward finalize: reduce the 2 or 3 partial slabs of one ln_bwd launch in ONE kernel
// (1024 threads = 64 columns x 16 partial phases; dw/db in the parameter dtype, dcol in f32).
template <typename OT>
__global__ __launch_bounds__(1024) void ln_bwd_finalize_kernel(const float* __restrict__ partials, int nblk, int D,
                                                                OT* __restrict__ dw, OT* __restrict__ db, float* __restrict__ dcol,
                                                                OT* __restrict__ dcol_t) {
    __shared__ float red[16][64];
    const int cx = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const int slab = blockIdx.y;
    const float* src = partials + (long long)slab * nblk * D;
    const float s = c < D ? fold_walk(src, D, nblk, ph, c) : 0.f;
    red[ph][cx] = s;
    __syncthreads();
    if (ph == 0 && c < D) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][cx];
        if (slab == 0) { if (dw) dw[c] = from_f32<OT>(t); }
        else if (slab == 1) { if (db) db[c] = from_f32<OT>(t); }
        else if (dcol) dcol[c] = t;
        else if (dcol_t) dcol_t[c] = from_f32<OT>(t);
    }
}

// stage 1 of a general column sum: block = 64 column-quads x 4 row phases over a 256-row slab.
constexpr int CS_ROWS = 256;
template <typename XT>
__global__ __launch_bounds__(256) void colsum_stage1_kernel(const XT* __restrict__ x, long long rows, long long cols,
                                                             long long ld, float* __restrict__ ws) {
    __shared__ f32x4 red[4][64];
    const int cx = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const long long c4 = ((long long)blockIdx.x * 64 + cx) * 4;
    const long long r0 = (long long)blockIdx.y * CS_ROWS;
    const long long r1 = r0 + CS_ROWS < rows ? r0 + CS_ROWS : rows;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (c4 < cols) {                // eight row loads in flight, added in row order (see fold_walk)
        long long r = r0 + ph;
        for (; r + 7 * 4 < r1; r += 8 * 4) {
            f32x4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = load4<XT>(x + (r + 4 * k) * ld + c4);
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k];
        }
        for (; r < r1; r += 4) s += load4<XT>(x + r * ld + c4);
    }
    red[ph][cx] = s;
    __syncthreads();
    if (ph == 0 && c4 < cols) {
        const f32x4 t = red[0][cx] + red[1][cx] + red[2][cx] + red[3][cx];
        *reinterpret_cast<f32x4*>(ws + (long long)blockIdx.y * cols + c4) = t;
    }
}

// scalar variant for widths that are not a multiple of 4 (e.g. num_classes = 10)
template <typename XT>
__global__ __launch_bounds__(256) void colsum_stage1_scalar_kernel(const XT* __restrict__ x, long long rows, long long cols,
                                                                    long long ld, float* __restrict__ ws) {
    __shared__ float red[4][64];
    const int cx = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const long long c = (long long)blockIdx.x * 64 + cx;
    const long long r0 = (long long)blockIdx.y * CS_ROWS;
    const long long r1 = r0 + CS_ROWS < rows ? r0 + CS_ROWS : rows;
    float s = 0.f;
    if (c < cols) for (long long r = r0 + ph; r < r1; r += 4) s += to_f32<XT>(x[r * ld + c]);
    red[ph][cx] = s;
    __syncthreads();
    if (ph == 0 && c < cols) ws[(long long)blockIdx.y * cols + c] = red[0][cx] + red[1][cx] + red[2][cx] + red[3][cx];
}

// ---- any row width (D % 4 != 0): element-at-a-time forms of the two kernels ------------------------------------------------
// T2T-ViT's token-to-token layers (t2t.py:45) are Transformers of width 3 * 7 * 7 = 147 and 147 * 9 = 1323.  Same maths and the
// same outputs (statistics, partial rows for vitk_layernorm_bwd_finalize) as the vector kernels; a lane owns the columns
// c = lane (mod 64) of every row its wave visits, so the column accumulators of the backward live in a per-wave LDS slab that
// only that lane touches (no atomics, fixed order).
template <typename XT, typename YT, typename WT>
__global__ __launch_bounds__(LN_THREADS) void ln_fwd_any_kernel(
    const XT* __restrict__ x, const WT* __restrict__ w, const WT* __restrict__ b, YT* __restrict__ y,
    float* __restrict__ mean_out, float* __restrict__ rstd_out, long long rows, int D, float eps, RowMap imap, RowMap omap,
    const WT* __restrict__ add, long long add_group, long long add_off) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float invD = 1.0f / (float)D;
    for (long long row = (long long)blockIdx.x * LN_WAVES + wave; row < rows; row += (long long)gridDim.x * LN_WAVES) {
        const XT* xr = x + map_row(imap, row) * (long long)D;
        float s = 0.f;
        for (int c = lane; c < D; c += 64) s += to_f32<XT>(xr[c]);
        const float mean = wave_sum(s) * invD;
        float q = 0.f;
        for (int c = lane; c < D; c += 64) { const float d = to_f32<XT>(xr[c]) - mean; q += d * d; }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * invD + eps);
        YT* yr = y + map_row(omap, row) * (long long)D;
        const WT* ar = add ? add + ((add_group > 0 ? row % add_group : row) + add_off) * (long long)D : nullptr;
        for (int c = lane; c < D; c += 64) {
            float o = (to_f32<XT>(xr[c]) - mean) * rstd * to_f32<WT>(w[c]) + (b ? to_f32<WT>(b[c]) : 0.f);
            if (ar) o += to_f32<WT>(ar[c]);
            yr[c] = from_f32<YT>(o);
        }
        if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
    }
}

template <typename DYT, typename XT, typename WT, typename DXT, int NW>
__global__ __launch_bounds__(NW * WAVE) void ln_bwd_any_kernel(
    const DYT* __restrict__ dy, const XT* __restrict__ x, const WT* __restrict__ w,
    const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
    const float* __restrict__ gin, float* __restrict__ dx_f32, DXT* __restrict__ dx_t,
    float* __restrict__ partials, int colsum_dx,
    long long rows, int D, RowMap dymap, RowMap xmap, RowMap dxmap, unsigned drop_t, unsigned drop_seed, float inv_keep) {
    extern __shared__ float acc_s[];                 // [NW][3][D]: dgamma, dbeta, colsum(dx_t) per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float invD = 1.0f / (float)D;
    const bool want_dx = (dx_f32 != nullptr) || (dx_t != nullptr);
    float* mine = acc_s + (long long)wave * 3 * D;
    for (int c = lane; c < 3 * D; c += 64) mine[c] = 0.f;
    for (long long row = (long long)blockIdx.x * NW + wave; row < rows; row += (long long)gridDim.x * NW) {
        const DYT* dyr = dy + map_row(dymap, row) * (long long)D;
        const XT* xr = x + map_row(xmap, row) * (long long)D;
        const float mean = mean_in[row], rstd = rstd_in[row];
        const long long orow = map_row(dxmap, row);
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < D; c += 64) {
            const float d = to_f32<DYT>(dyr[c]);
            const float xh = (to_f32<XT>(xr[c]) - mean) * rstd;
            const float g = d * to_f32<WT>(w[c]);
            s1 += g; s2 += g * xh;
            mine[c] += d * xh;
            mine[D + c] += d;
        }
        if (want_dx) {
            const float c1 = wave_sum(s1) * invD, c2 = wave_sum(s2) * invD;
            const unsigned hrow = drop_row((unsigned)orow, drop_seed);
            for (int c = lane; c < D; c += 64) {
                const float xh = (to_f32<XT>(xr[c]) - mean) * rstd;
                const float g = to_f32<DYT>(dyr[c]) * to_f32<WT>(w[c]);
                float o = rstd * (g - c1 - xh * c2);
                if (gin) o += gin[orow * D + c];
                if (dx_f32) dx_f32[orow * D + c] = o;
                if (drop_t) o = drop_keep(hrow, (unsigned)c, drop_t) ? o * inv_keep : 0.f;      // as in ln_bwd_kernel: dx_t / its sums only
                if (dx_t) dx_t[orow * D + c] = from_f32<DXT>(o);
                if (colsum_dx) mine[2 * D + c] += o;
            }
        }
    }
    __syncthreads();
    const int nslab = colsum_dx ? 3 : 2;
    for (int i = threadIdx.x; i < nslab * D; i += NW * WAVE) {
        const int slab = i / D, c = i - slab * D;
        float sum = 0.f;
#pragma unroll
        for (int wv_ = 0; wv_ < NW; ++wv_) sum += acc_s[((long long)wv_ * 3 + slab) * D + c];
        partials[((long long)slab * gridDim.x + blockIdx.x) * D + c] = sum;
    }
}

template <typename XT, typename YT, typename WT>
int launch_ln_fwd(const void* x, const void* w, const void* b, void* y, float* mean, float* rstd, long long rows, int D,
                  float eps, RowMap im, RowMap om, const void* add, long long ag, long long ao, hipStream_t st, F8Out f8) {
    const int nchunk = D / 4;
    const int maxc = (nchunk + 63) / 64;
    long long blocks = (rows + LN_WAVES - 1) / LN_WAVES;
    if (blocks > 8192) blocks = 8192;
    if (f8.amax && blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    if (D & 3) {
        hipLaunchKernelGGL((ln_fwd_any_kernel<XT, YT, WT>), dim3((unsigned)blocks), dim3(LN_THREADS), 0, st, (const XT*)x, (const WT*)w,
                           (const WT*)b, (YT*)y, mean, rstd, rows, D, eps, im, om, (const WT*)add, ag, ao);
        VITK_CHECK_LAUNCH("layernorm_fwd");
        return 0;
    }
    // 16-bit rows of 768 / 1024 / 1280 elements with identity maps and nothing fused: two rows per wave, 16-byte accesses (ln_fwd16_kernel);
    // VITK_LN_FWD16=0 keeps the general kernel (A/B)
    if constexpr (std::is_same<XT, __bf16>::value && std::is_same<YT, __bf16>::value && std::is_same<WT, __bf16>::value) {
        static const bool off16 = vitk_exp("VITK_LN_FWD16") && atoi(vitk_exp("VITK_LN_FWD16")) == 0;
        const int cpl = D / 256;
        if (!off16 && D % 256 == 0 && cpl >= 3 && cpl <= 5 && rows >= 1024 && im.group <= 0 && om.group <= 0 && !add && !f8.p && !f8.amax && aligned16(w) && (!b || aligned16(b))) {
#define LN_FWD16_LAUNCH(CPL_, HB_) do { \
                static const long long resident = [] { int per_cu = 0, dev = 0, cus = 0; \
                    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ln_fwd16_kernel<CPL_, HB_>, LN_THREADS, 0) != hipSuccess || per_cu < 1) per_cu = 4; \
                    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256; \
                    return (long long)per_cu * cus; }(); \
                long long nb = (rows + 2 * LN_WAVES - 1) / (2 * LN_WAVES); \
                if (nb > resident) nb = resident; \
                hipLaunchKernelGGL((ln_fwd16_kernel<CPL_, HB_>), dim3((unsigned)nb), dim3(LN_THREADS), 0, st, (const __bf16*)x, (const __bf16*)w, (const __bf16*)b, \
                                   (__bf16*)y, mean, rstd, rows, eps); } while (0)
#define LN_FWD16_CASE(CPL_) do { if (b) LN_FWD16_LAUNCH(CPL_, true); else LN_FWD16_LAUNCH(CPL_, false); } while (0)
            if (cpl == 3) LN_FWD16_CASE(3); else if (cpl == 4) LN_FWD16_CASE(4); else LN_FWD16_CASE(5);
#undef LN_FWD16_CASE
#undef LN_FWD16_LAUNCH
            VITK_CHECK_LAUNCH("layernorm_fwd (16-bit rows)");
            return 0;
        }
    }
    // nontemporal row loads (VITK_LN_FWD_NT=1): [measured, round 3] no gain here, unlike the backward -- two interleaved bench runs
    // 39.2-39.5 ms with the hint vs 39.2-39.3 without (the forward's rows are re-read soon, by the residual epilogue); left opt-in
    const char* nt_env = vitk_exp("VITK_LN_FWD_NT");
    const bool nt = rows >= 4096 && nt_env && nt_env[0] == '1';
    // grid = the blocks that are resident at once (occupancy x CUs, asked once per instantiation): every wave then walks its rows with
    // the next one in flight; VITK_LN_FWD_BLOCKS overrides the cap
#define LN_FWD_LAUNCH(KERNEL) do { \
        static const long long resident = [] { int per_cu = 0, dev = 0, cus = 0; \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, KERNEL, LN_THREADS, 0) != hipSuccess || per_cu < 1) per_cu = 4; \
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256; \
            return (long long)per_cu * cus; }(); \
        long long cap = vitk_exp("VITK_LN_FWD_BLOCKS") ? atoll(vitk_exp("VITK_LN_FWD_BLOCKS")) : resident; \
        if (cap < 1) cap = 1; \
        const long long nb = blocks < cap ? blocks : cap; \
        hipLaunchKernelGGL(KERNEL, dim3((unsigned)nb), dim3(LN_THREADS), 0, st, (const XT*)x, (const WT*)w, (const WT*)b, (YT*)y, mean, rstd, \
                           rows, D, eps, im, om, (const WT*)add, ag, ao, f8); } while (0)
#define LN_FWD_CASE(MC) do { if (nt) LN_FWD_LAUNCH((ln_fwd_kernel<XT, YT, WT, MC, true>)); \
                             else LN_FWD_LAUNCH((ln_fwd_kernel<XT, YT, WT, MC, false>)); } while (0)
    if (maxc <= 1) LN_FWD_CASE(1);
    else if (maxc <= 3) LN_FWD_CASE(3);
    else if (maxc <= 4) LN_FWD_CASE(4);
    else if (maxc <= 5) LN_FWD_CASE(5);
    else if (maxc <= 8) LN_FWD_CASE(8);
    else LN_FWD_CASE(16);
#undef LN_FWD_CASE
#undef LN_FWD_LAUNCH
    VITK_CHECK_LAUNCH("layernorm_fwd");
    return 0;
}

template <typename DYT, typename XT, typename WT, typename DXT, typename GT = float>
int launch_ln_bwd(const void* dy, const void* x, const void* w, const float* mean, const float* rstd, const void* gin_,
                  float* dxf, void* dxt, float* partials, int colsum_dx, long long rows, int D, RowMap dm, RowMap xm,
                  RowMap om, hipStream_t st, unsigned drop_t, unsigned drop_seed, float inv_keep) {
    const GT* gin = (const GT*)gin_;
    constexpr bool F32_STREAM = std::is_same<GT, float>::value;
    const int nchunk = D / 4;
    const int maxc = (nchunk + 63) / 64;
    const long long blocks = vitk_layernorm_bwd_blocks(rows, D);
    if (D & 3) {
        if constexpr (!F32_STREAM) VITK_FAIL(VITK_E_SHAPE, "layernorm_bwd_s16: the 16-bit stream needs D %% 4 == 0 (got %d)", D);
        else {
        const bool wide = (D / 4 + 63) / 64 >= 4;          // the block shape vitk_layernorm_bwd_blocks assumed
        const size_t lds = (size_t)(wide ? 4 : 8) * 3 * D * sizeof(float);
        if (lds > 150 * 1024) VITK_FAIL(VITK_E_SHAPE, "layernorm_bwd: widths that are not multiples of 4 are served up to D = 3200 (got %d)", D);
#define LN_BWD_ANY(NWV) do { \
            static const int rc__ = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(ln_bwd_any_kernel<DYT, XT, WT, DXT, NWV>), \
                                                             hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); \
            if (rc__ != 0) VITK_FAIL(rc__, "layernorm_bwd: cannot enable 150 KiB of LDS"); \
            hipLaunchKernelGGL((ln_bwd_any_kernel<DYT, XT, WT, DXT, NWV>), dim3((unsigned)blocks), dim3(NWV * WAVE), lds, st, (const DYT*)dy, \
                               (const XT*)x, (const WT*)w, mean, rstd, gin, dxf, (DXT*)dxt, partials, colsum_dx, rows, D, dm, xm, om, drop_t, drop_seed, inv_keep); \
        } while (0)
        if (wide) LN_BWD_ANY(4); else LN_BWD_ANY(8);
#undef LN_BWD_ANY
        VITK_CHECK_LAUNCH("layernorm_bwd");
        return 0;
        }
    }
    static const int no_fast = vitk_exp("VITK_LNB_FAST") ? !atoi(vitk_exp("VITK_LNB_FAST")) : 0;
    if (!no_fast && gin && (dxf || !F32_STREAM) && dxt && colsum_dx && dm.group <= 0 && xm.group <= 0 && om.group <= 0 && (D == 768 || D == 1024 || D == 1280)) {
        // nontemporal loads of the three row streams: 123 -> 99 us at 50432 x 768 (6.3 TB/s); prefetching the next row (PIPE) on
        // top of them costs 10 us, so it stays a switch
#define LN_BWD_FAST(MC, NWV, WPE, NTL, PIPE, DROP) hipLaunchKernelGGL((ln_bwd_fast_kernel<DYT, XT, WT, DXT, GT, MC, NWV, WPE, NTL, PIPE, DROP>), \
        dim3((unsigned)blocks), dim3(NWV * WAVE), 0, st, \
        (const DYT*)dy, (const XT*)x, (const WT*)w, mean, rstd, gin, dxf, (DXT*)dxt, partials, rows, drop_t, drop_seed, inv_keep)
        static const int fnt = vitk_exp("VITK_LNB_NT") ? atoi(vitk_exp("VITK_LNB_NT")) : 1;
        static const int fpipe = vitk_exp("VITK_LNB_PIPE") ? atoi(vitk_exp("VITK_LNB_PIPE")) : 0;
#define LN_BWD_FAST_D(MC, NWV, WPE) \
        if (drop_t) LN_BWD_FAST(MC, NWV, WPE, true, false, true); \
        else if (fnt && fpipe) LN_BWD_FAST(MC, NWV, WPE, true, true, false); \
        else if (fnt) LN_BWD_FAST(MC, NWV, WPE, true, false, false); \
        else if (fpipe) LN_BWD_FAST(MC, NWV, WPE, false, true, false); \
        else LN_BWD_FAST(MC, NWV, WPE, false, false, false)
        if (D == 768) { LN_BWD_FAST_D(3, 8, 4); }
        else if (D == 1024) { LN_BWD_FAST_D(4, 4, 3); }
        else { LN_BWD_FAST_D(5, 4, 2); }
#undef LN_BWD_FAST_D
#undef LN_BWD_FAST
        VITK_CHECK_LAUNCH("layernorm_bwd");
        return 0;
    }
#define LN_BWD_CASE(MC) hipLaunchKernelGGL((ln_bwd_kernel<DYT, XT, WT, DXT, GT, MC, (MC >= 4 ? 4 : 8)>), dim3((unsigned)blocks), dim3((MC >= 4 ? 4 : 8) * WAVE), 0, st, \
        (const DYT*)dy, (const XT*)x, (const WT*)w, mean, rstd, gin, dxf, (DXT*)dxt, partials, colsum_dx, rows, D, dm, xm, om, drop_t, drop_seed, inv_keep)
    if (maxc <= 1) LN_BWD_CASE(1);
    else if (maxc <= 3) LN_BWD_CASE(3);
    else if (maxc <= 4) LN_BWD_CASE(4);
    else if (maxc <= 5) LN_BWD_CASE(5);
    else if (maxc <= 8) LN_BWD_CASE(8);
    else LN_BWD_CASE(16);
#undef LN_BWD_CASE
    VITK_CHECK_LAUNCH("layernorm_bwd");
    return 0;
}

}  // namespace

extern "C" int vitk_layernorm_fwd(const void* x, int xdt, const void* w, const void* b, int wdt, void* y, int ydt,
                                  float* mean, float* rstd, int64_t rows, int64_t D, float eps, vitk_rowmap imap,
                                  vitk_rowmap omap, const void* add, int64_t add_group, int64_t add_off, void* stream) {
    return vitk_layernorm_fwd_fp8(x, xdt, w, b, wdt, y, ydt, mean, rstd, rows, D, eps, imap, omap, add, add_group, add_off, nullptr,
                                  nullptr, nullptr, stream);
}

extern "C" int vitk_layernorm_fwd_fp8(const void* x, int xdt, const void* w, const void* b, int wdt, void* y, int ydt,
                                      float* mean, float* rstd, int64_t rows, int64_t D, float eps, vitk_rowmap imap,
                                      vitk_rowmap omap, const void* add, int64_t add_group, int64_t add_off, void* y8,
                                      const float* scale8, uint32_t* amax64, void* stream) {
    if (y8 && !scale8) VITK_FAIL(VITK_E_ARG, "layernorm_fwd_fp8: an fp8 output needs its scale");
    const F8Out f8{(unsigned char*)y8, scale8, (unsigned*)amax64};
    if (!x || !w || !y || !mean || !rstd) VITK_FAIL(VITK_E_ARG, "layernorm_fwd: null pointer");
    if (rows < 0 || D <= 0 || D > 4096) VITK_FAIL(VITK_E_SHAPE, "layernorm_fwd: need 0 < D <= 4096, got D=%lld", (long long)D);
    if ((D & 3) && (y8 || amax64)) VITK_FAIL(VITK_E_SHAPE, "layernorm_fwd_fp8: the fp8 side output needs D %% 4 == 0");
    if (rows == 0) return 0;
    if (!(D & 3) && (!aligned16(x) || !aligned16(y) || !aligned8(w) || (b && !aligned8(b)) || (add && !aligned8(add))))
        VITK_FAIL(VITK_E_ALIGN, "layernorm_fwd: pointers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const RowMap im = to_map(imap), om = to_map(omap);
    if (wdt == VITK_F32) {
        if (xdt != VITK_F32 || ydt != VITK_F32) VITK_FAIL(VITK_E_DTYPE, "layernorm_fwd: f32 params need f32 x/y");
        return launch_ln_fwd<float, float, float>(x, w, b, y, mean, rstd, rows, (int)D, eps, im, om, add, add_group, add_off, st, f8);
    }
    if (wdt != VITK_BF16) VITK_FAIL(VITK_E_DTYPE, "layernorm_fwd: bad wdt");
    if (xdt == VITK_F32 && ydt == VITK_BF16) return launch_ln_fwd<float, __bf16, __bf16>(x, w, b, y, mean, rstd, rows, (int)D, eps, im, om, add, add_group, add_off, st, f8);
    if (xdt == VITK_F32 && ydt == VITK_F32) return launch_ln_fwd<float, float, __bf16>(x, w, b, y, mean, rstd, rows, (int)D, eps, im, om, add, add_group, add_off, st, f8);
    if (xdt == VITK_BF16 && ydt == VITK_BF16) return launch_ln_fwd<__bf16, __bf16, __bf16>(x, w, b, y, mean, rstd, rows, (int)D, eps, im, om, add, add_group, add_off, st, f8);
    if (xdt == VITK_BF16 && ydt == VITK_F32) return launch_ln_fwd<__bf16, float, __bf16>(x, w, b, y, mean, rstd, rows, (int)D, eps, im, om, add, add_group, add_off, st, f8);
    VITK_FAIL(VITK_E_DTYPE, "layernorm_fwd: bad dtype combination");
}

// ---- fused patch gather + LayerNorm(patch_dim) (see patch_ln_fwd16_kernel) ----
extern "C" int vitk_patch_ln_serves(int dt, int64_t C, int64_t H, int64_t W, int64_t p1, int64_t p2) {
    return dt == VITK_BF16 && C == 3 && p1 == 16 && p2 == 16 && H > 0 && W > 0 && H % 16 == 0 && W % 16 == 0 && !vitk_exp("VITK_NO_PATCH_LN");
}
extern "C" int64_t vitk_patch_ln_bwd_blocks(int64_t rows) {
    int64_t nb = (rows + 2 * LN_WAVES - 1) / (2 * LN_WAVES);
    return nb > 512 ? 512 : (nb < 1 ? 1 : nb);
}
extern "C" int vitk_patch_ln_fwd(const void* img, int dt, const void* w, const void* b, void* y, float* mean, float* rstd, int64_t B,
                                 int64_t C, int64_t H, int64_t W, int64_t p1, int64_t p2, float eps, void* stream) {
    if (!img || !w || !y || !mean || !rstd) VITK_FAIL(VITK_E_ARG, "patch_ln_fwd: null pointer");
    if (!vitk_patch_ln_serves(dt, C, H, W, p1, p2) || B <= 0) VITK_FAIL(VITK_E_SHAPE, "patch_ln_fwd: serves 16-bit images of 3 channels with 16 x 16 patches (ask vitk_patch_ln_serves)");
    if (!aligned16(img) || !aligned16(w) || (b && !aligned16(b)) || !aligned16(y)) VITK_FAIL(VITK_E_ALIGN, "patch_ln_fwd: pointers must be 16-byte aligned");
    const int hp = (int)(H / 16), wp = (int)(W / 16);
    const long long rows = (long long)B * hp * wp;
    long long nb = (rows + 2 * LN_WAVES - 1) / (2 * LN_WAVES);
    if (nb > 4096) nb = 4096;
    hipStream_t st = (hipStream_t)stream;
    if (b) hipLaunchKernelGGL((patch_ln_fwd16_kernel<true>), dim3((unsigned)nb), dim3(LN_THREADS), 0, st, (const __bf16*)img, (const __bf16*)w, (const __bf16*)b,
                              (__bf16*)y, mean, rstd, rows, hp, wp, (long long)H * W, (int)W, eps);
    else hipLaunchKernelGGL((patch_ln_fwd16_kernel<false>), dim3((unsigned)nb), dim3(LN_THREADS), 0, st, (const __bf16*)img, (const __bf16*)w, (const __bf16*)nullptr,
                            (__bf16*)y, mean, rstd, rows, hp, wp, (long long)H * W, (int)W, eps);
    VITK_CHECK_LAUNCH("patch_ln_fwd");
    return 0;
}
extern "C" int vitk_patch_ln_bwd_params(const void* dy, const void* img, int dt, const float* mean, const float* rstd, float* partials,
                                        int64_t B, int64_t C, int64_t H, int64_t W, int64_t p1, int64_t p2, void* stream) {
    if (!dy || !img || !mean || !rstd || !partials) VITK_FAIL(VITK_E_ARG, "patch_ln_bwd_params: null pointer");
    if (!vitk_patch_ln_serves(dt, C, H, W, p1, p2) || B <= 0) VITK_FAIL(VITK_E_SHAPE, "patch_ln_bwd_params: serves 16-bit images of 3 channels with 16 x 16 patches");
    if (!aligned16(img) || !aligned16(dy)) VITK_FAIL(VITK_E_ALIGN, "patch_ln_bwd_params: pointers must be 16-byte aligned");
    const int hp = (int)(H / 16), wp = (int)(W / 16);
    const long long rows = (long long)B * hp * wp;
    hipLaunchKernelGGL(patch_ln_bwd16_kernel, dim3((unsigned)vitk_patch_ln_bwd_blocks(rows)), dim3(LN_THREADS), 0, (hipStream_t)stream, (const __bf16*)dy,
                       (const __bf16*)img, mean, rstd, partials, rows, hp, wp, (long long)H * W, (int)W);
    VITK_CHECK_LAUNCH("patch_ln_bwd_params");
    return 0;
}

extern "C" int64_t vitk_layernorm_bwd_blocks(int64_t rows, int64_t D) {
    // one partial row per block; the block shape follows the row width (see ln_bwd_kernel): 8 waves and up to 512
    // blocks (4 per CU) up to 768 columns, 4 waves and up to 768 blocks (3 per CU, all resident at once) beyond
    const bool wide = (D / 4 + 63) / 64 >= 4;
    const int64_t nw = wide ? 4 : LNB_WAVES, cap = wide ? 768 : LNB_MAX_BLOCKS;
    int64_t blocks = (rows + nw - 1) / nw;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return blocks;
}

extern "C" int vitk_layernorm_bwd(const void* dy, int dydt, const void* x, int xdt, const void* w, int wdt,
                                  const float* mean, const float* rstd, const float* gin, float* dx_f32, void* dx_t,
                                  int dxtdt, float* partials, int colsum_dx, int64_t rows, int64_t D, vitk_rowmap dymap,
                                  vitk_rowmap xmap, vitk_rowmap dxmap, void* stream) {
    return vitk_layernorm_bwd_drop(dy, dydt, x, xdt, w, wdt, mean, rstd, gin, dx_f32, dx_t, dxtdt, partials, colsum_dx, rows, D, dymap,
                                   xmap, dxmap, 0.f, 0u, stream);
}

extern "C" int vitk_layernorm_bwd_drop(const void* dy, int dydt, const void* x, int xdt, const void* w, int wdt,
                                       const float* mean, const float* rstd, const float* gin, float* dx_f32, void* dx_t,
                                       int dxtdt, float* partials, int colsum_dx, int64_t rows, int64_t D, vitk_rowmap dymap,
                                       vitk_rowmap xmap, vitk_rowmap dxmap, float drop_p, uint32_t drop_seed, void* stream) {
    if (!dy || !x || !w || !mean || !rstd || !partials) VITK_FAIL(VITK_E_ARG, "layernorm_bwd: null pointer");
    if (!(drop_p >= 0.f && drop_p < 1.f)) VITK_FAIL(VITK_E_ARG, "layernorm_bwd: dropout p must be in [0, 1) (got %g)", (double)drop_p);
    const unsigned drop_t = drop_thresh(drop_p);
    const float inv_keep = 1.0f / (1.0f - drop_p);
    if (rows <= 0 || D <= 0 || D > 4096) VITK_FAIL(VITK_E_SHAPE, "layernorm_bwd: need rows > 0, 0 < D <= 4096");
    if (!(D & 3) && (!aligned16(dy) || !aligned16(x) || (gin && !aligned16(gin)) || (dx_f32 && !aligned16(dx_f32)) || (dx_t && !aligned16(dx_t))))
        VITK_FAIL(VITK_E_ALIGN, "layernorm_bwd: pointers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const RowMap dm = to_map(dymap), xm = to_map(xmap), om = to_map(dxmap);
    if (wdt == VITK_F32) {
        if (dydt != VITK_F32 || xdt != VITK_F32 || (dx_t && dxtdt != VITK_F32)) VITK_FAIL(VITK_E_DTYPE, "layernorm_bwd: f32 params need f32 tensors");
        return launch_ln_bwd<float, float, float, float>(dy, x, w, mean, rstd, gin, dx_f32, dx_t, partials, colsum_dx, rows, (int)D, dm, xm, om, st, drop_t, drop_seed, inv_keep);
    }
    if (wdt != VITK_BF16) VITK_FAIL(VITK_E_DTYPE, "layernorm_bwd: bad wdt");
    if (dx_t && dxtdt != VITK_BF16) VITK_FAIL(VITK_E_DTYPE, "layernorm_bwd: dx_t must be bf16 with bf16 params");
    if (dydt == VITK_BF16 && xdt == VITK_F32) return launch_ln_bwd<__bf16, float, __bf16, __bf16>(dy, x, w, mean, rstd, gin, dx_f32, dx_t, partials, colsum_dx, rows, (int)D, dm, xm, om, st, drop_t, drop_seed, inv_keep);
    if (dydt == VITK_BF16 && xdt == VITK_BF16) return launch_ln_bwd<__bf16, __bf16, __bf16, __bf16>(dy, x, w, mean, rstd, gin, dx_f32, dx_t, partials, colsum_dx, rows, (int)D, dm, xm, om, st, drop_t, drop_seed, inv_keep);
    if (dydt == VITK_F32 && xdt == VITK_F32) return launch_ln_bwd<float, float, __bf16, __bf16>(dy, x, w, mean, rstd, gin, dx_f32, dx_t, partials, colsum_dx, rows, (int)D, dm, xm, om, st, drop_t, drop_seed, inv_keep);
    if (dydt == VITK_F32 && xdt == VITK_BF16) return launch_ln_bwd<float, __bf16, __bf16, __bf16>(dy, x, w, mean, rstd, gin, dx_f32, dx_t, partials, colsum_dx, rows, (int)D, dm, xm, om, st, drop_t, drop_seed, inv_keep);
    VITK_FAIL(VITK_E_DTYPE, "layernorm_bwd: bad dtype combination");
}

extern "C" int vitk_layernorm_bwd_s16(const void* dy, int dydt, const void* x, int xdt, const void* w, int wdt, const float* mean,
                                      const float* rstd, const void* gin, void* dx_t, int dxtdt, float* partials, int colsum_dx,
                                      int64_t rows, int64_t D, vitk_rowmap dymap, vitk_rowmap xmap, vitk_rowmap dxmap, void* stream) {
    if (!dy || !x || !w || !mean || !rstd || !partials || !dx_t) VITK_FAIL(VITK_E_ARG, "layernorm_bwd_s16: null pointer");
    if (rows <= 0 || D <= 0 || D > 4096 || (D & 3)) VITK_FAIL(VITK_E_SHAPE, "layernorm_bwd_s16: need rows > 0, 0 < D <= 4096, D %% 4 == 0");
    if (!aligned16(dy) || !aligned16(x) || (gin && !aligned16(gin)) || !aligned16(dx_t)) VITK_FAIL(VITK_E_ALIGN, "layernorm_bwd_s16: pointers must be 16-byte aligned");
    if (wdt != VITK_BF16 || dxtdt != VITK_BF16) VITK_FAIL(VITK_E_DTYPE, "layernorm_bwd_s16: the 16-bit stream comes with 16-bit parameters");
    hipStream_t st = (hipStream_t)stream;
    const RowMap dm = to_map(dymap), xm = to_map(xmap), om = to_map(dxmap);
    if (dydt == VITK_BF16 && xdt == VITK_F32) return launch_ln_bwd<__bf16, float, __bf16, __bf16, __bf16>(dy, x, w, mean, rstd, gin, nullptr, dx_t, partials, colsum_dx, rows, (int)D, dm, xm, om, st, 0u, 0u, 1.0f);
    if (dydt == VITK_BF16 && xdt == VITK_BF16) return launch_ln_bwd<__bf16, __bf16, __bf16, __bf16, __bf16>(dy, x, w, mean, rstd, gin, nullptr, dx_t, partials, colsum_dx, rows, (int)D, dm, xm, om, st, 0u, 0u, 1.0f);
    VITK_FAIL(VITK_E_DTYPE, "layernorm_bwd_s16: dy must be 16-bit, x float32 or 16-bit");
}

extern "C" int vitk_layernorm_bwd_finalize(const float* partials, int64_t nblk, int64_t D, void* dw, void* db, int odt,
                                           float* dcol, void* stream) {
    return vitk_layernorm_bwd_finalize_ex(partials, nblk, D, dw, db, odt, dcol, VITK_F32, stream);
}

extern "C" int vitk_layernorm_bwd_finalize_ex(const float* partials, int64_t nblk, int64_t D, void* dw, void* db, int odt,
                                              void* dcol, int dcol_dt, void* stream) {
    if (!partials) VITK_FAIL(VITK_E_ARG, "layernorm_bwd_finalize: null pointer");
    if (nblk <= 0 || D <= 0) VITK_FAIL(VITK_E_SHAPE, "layernorm_bwd_finalize: empty");
    if (dcol && dcol_dt != VITK_F32 && dcol_dt != odt) VITK_FAIL(VITK_E_DTYPE, "layernorm_bwd_finalize: dcol is float32 or of the parameter dtype");
    const dim3 grid((unsigned)((D + 63) / 64), dcol ? 3u : 2u);
    float* dcol_f = (dcol && dcol_dt == VITK_F32) ? (float*)dcol : nullptr;
    void* dcol_t = (dcol && dcol_dt != VITK_F32) ? dcol : nullptr;
    VITK_DISPATCH_DT(odt, OT, hipLaunchKernelGGL((ln_bwd_finalize_kernel<OT>), grid, dim3(1024), 0, (hipStream_t)stream, partials,
                                                  (int)nblk, (int)D, (OT*)dw, (OT*)db, dcol_f, (OT*)dcol_t));
    VITK_CHECK_LAUNCH("layernorm_bwd_finalize");
    return 0;
}

extern "C" int vitk_colsum_partials(const float* partials, int64_t nparts, int64_t ld, int64_t cols, void* out, int odt,
                                    int accumulate, void* stream) {
    if (!partials || !out) VITK_FAIL(VITK_E_ARG, "colsum_partials: null pointer");
    if (cols <= 0 || nparts <= 0 || nparts > 0x7fffffff) VITK_FAIL(VITK_E_SHAPE, "colsum_partials: empty (or more than 2^31 partial rows)");
    const unsigned blocks = (unsigned)((cols + 63) / 64);
    VITK_DISPATCH_DT(odt, OT, hipLaunchKernelGGL((colsum_partials_kernel<OT>), dim3(blocks), dim3(1024), 0, (hipStream_t)stream,
                                                  partials, (long long)nparts, (long long)ld, (long long)cols, (OT*)out, accumulate));
    VITK_CHECK_LAUNCH("colsum_partials");
    return 0;
}

extern "C" int vitk_fold_many(const float* const* src, void* const* dst, const int64_t* nparts, const int64_t* ld, const int64_t* cols,
                              const int32_t* flags, int64_t count, void* stream) {
    if (count <= 0) return 0;
    if (!src || !dst || !nparts || !ld || !cols || !flags) VITK_FAIL(VITK_E_ARG, "fold_many: null table");
    hipStream_t st = (hipStream_t)stream;
    int64_t t = 0;
    while (t < count) {
        FoldMany a;
        a.count = 0; a.blk0[0] = 0;
        while (t < count && a.count < FM_MAX) {
            if (!src[t] || !dst[t]) VITK_FAIL(VITK_E_ARG, "fold_many: null pointer in the table");
            if (nparts[t] <= 0 || cols[t] <= 0 || ld[t] < cols[t] || nparts[t] > 0x7fffffff || ld[t] > 0x7fffffff) VITK_FAIL(VITK_E_SHAPE, "fold_many: bad job %lld", (long long)t);
            const int dt_ = flags[t] >> 4;
            if ((flags[t] & ~0x31) || (dt_ != VITK_F32 && dt_ != VITK_BF16)) VITK_FAIL(VITK_E_DTYPE, "fold_many: bad flags / dtype tag in job %lld", (long long)t);
            const int k = a.count;
            a.src[k] = src[t]; a.dst[k] = dst[t]; a.nparts[k] = (int)nparts[t]; a.ld[k] = (int)ld[t]; a.cols[k] = (int)cols[t]; a.flags[k] = flags[t];
            a.blk0[k + 1] = a.blk0[k] + (int)((cols[t] + 63) / 64);
            ++a.count; ++t;
        }
        hipLaunchKernelGGL(fold_many_kernel, dim3((unsigned)a.blk0[a.count]), dim3(1024), 0, st, a);
        VITK_CHECK_LAUNCH("fold_many");
    }
    return 0;
}

extern "C" int64_t vitk_colsum_ws_floats(int64_t rows, int64_t cols) { return ((rows + CS_ROWS - 1) / CS_ROWS) * cols; }

extern "C" int vitk_colsum(const void* x, int xdt, int64_t rows, int64_t cols, int64_t ld, void* out, int odt,
                           int accumulate, float* ws, void* stream) {
    if (!x || !out || !ws) VITK_FAIL(VITK_E_ARG, "colsum: null pointer");
    if (rows <= 0 || cols <= 0 || ld < cols) VITK_FAIL(VITK_E_SHAPE, "colsum: bad shape");
    const long long rb = (rows + CS_ROWS - 1) / CS_ROWS;
    if (rb > 65535) VITK_FAIL(VITK_E_SHAPE, "colsum: too many rows");
    if ((cols & 3) == 0 && (ld & 3) == 0 && aligned16(x) && aligned16(ws)) {
        const dim3 grid((unsigned)((cols / 4 + 63) / 64), (unsigned)rb);
        VITK_DISPATCH_DT(xdt, XT, hipLaunchKernelGGL((colsum_stage1_kernel<XT>), grid, dim3(256), 0, (hipStream_t)stream,
                                                      (const XT*)x, (long long)rows, (long long)cols, (long long)ld, ws));
    } else {
        const dim3 grid((unsigned)((cols + 63) / 64), (unsigned)rb);
        VITK_DISPATCH_DT(xdt, XT, hipLaunchKernelGGL((colsum_stage1_scalar_kernel<XT>), grid, dim3(256), 0, (hipStream_t)stream,
                                                      (const XT*)x, (long long)rows, (long long)cols, (long long)ld, ws));
    }
    VITK_CHECK_LAUNCH("colsum_stage1");
    return vitk_colsum_partials(ws, rb, cols, cols, out, odt, accumulate, stream);
}
