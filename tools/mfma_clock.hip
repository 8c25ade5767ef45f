// Sustained MFMA rate and shader clock under matrix load: hipcc --offload-arch=gfx950 -O3 tools/mfma_clock.hip -o /tmp/mfma_clock
// Every wave issues ITER x 16 independent v_mfma_f32_16x16x32_bf16 from registers (no memory traffic); s_memtime
// (shader clock) against wall_clock64 (100 MHz) gives the clock the matrix pipes actually sustain.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* clk, int iters) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(e * 0.5f); }
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}
int main() {
    for (int waves = 4; waves <= 8; waves += 4) {
        const int blocks = 256, threads = 64 * waves, iters = 20000;
        float* out; unsigned long long* clk;
        hipMalloc(&out, blocks * threads * 4); hipMalloc(&clk, blocks * 16);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<<<blocks, threads>>>(out, clk, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0); k<<<blocks, threads>>>(out, clk, iters); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[512]; hipMemcpy(h, clk, blocks * 16, hipMemcpyDeviceToHost);
        double cyc = 0, wall = 0; for (int i = 0; i < blocks; ++i) { cyc += h[2 * i]; wall += h[2 * i + 1]; }
        cyc /= blocks; wall /= blocks;
        const double flops = (double)blocks * waves * iters * 16 * 16 * 16 * 32 * 2;
        printf("waves/CU %d: %.3f ms, %.1f TFLOP/s; per-WG %.0f shader cycles in %.1f us -> %.3f GHz; %.2f cycles per MFMA per SIMD\n",
               waves, ms, flops / ms / 1e9, cyc, wall / 100.0, cyc / (wall / 100.0) / 1e3, cyc / ((double)iters * 16 * waves / 4));
    }
    return 0;
}
