// Does gfx950 execute scalar atomics (s_atomic_add ... glc, returned value in an SGPR, tracked by lgkmcnt)?
// hipcc --offload-arch=gfx950 -O2 tools/probe_satomic.hip -o tools/probe_satomic.bin && tools/probe_satomic.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* ctr, unsigned* out) {
    unsigned v = 1;
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(ctr) : "memory");
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}
int main() {
    unsigned *ctr, *out;
    hipMalloc(&ctr, 4); hipMalloc(&out, 4 * 1024);
    hipMemset(ctr, 0, 4);
    hipLaunchKernelGGL(k, dim3(1024), dim3(64), 0, 0, ctr, out);
    if (hipDeviceSynchronize() != hipSuccess) { printf("FAULT\n"); return 1; }
    unsigned h[1024], c;
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(&c, ctr, 4, hipMemcpyDeviceToHost);
    unsigned long long sum = 0; unsigned mx = 0;
    for (int i = 0; i < 1024; ++i) { sum += h[i]; if (h[i] > mx) mx = h[i]; }
    printf("counter=%u (expect 1024) sum of returns=%llu (expect %llu) max=%u (expect 1023)\n", c, sum, 1023ull * 1024 / 2, mx);
    return 0;
}
