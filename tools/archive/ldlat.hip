// Micro-benchmark: how fast can a CU stream GEMM-like operand tiles into LDS by LDS-DMA as a function of the number of K-steps
// in flight?  Per K-step every workgroup (one per CU) loads 16 KiB of a "W" panel that ALL workgroups read at the same time
// (L2-hot) and 16 KiB of an "A" panel that `share` workgroups of the same XCD read at the same time and nobody read before
// (a compulsory miss for the first toucher: A panels tile a 160 MiB region).  DEPTH stages of 32 KiB, DEPTH - 1 in flight.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int DEPTH>
__global__ __launch_bounds__(512) void fill(const char* __restrict__ a, const char* __restrict__ w, int iters, int share, long long a_region, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
    const long long a_base = ((long long)(xcd * 32 + l / share) * iters * 16384) % a_region;     // private to `share` neighbours
    int acc = 0;
    for (int it = 0; it < iters; ++it) {
        char* st = lds + (it % DEPTH) * 32768;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(a + a_base + (long long)it * 16384 + (wave * 2 + j) * 1024 + lane * 16),
                                             (void __attribute__((address_space(3)))*)(st + (wave * 2 + j) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(w + ((long long)it * 16384) % (2 << 20) + (wave * 2 + j) * 1024 + lane * 16),
                                             (void __attribute__((address_space(3)))*)(st + 16384 + (wave * 2 + j) * 1024), 16, 0, 0);
        }
        if (it >= DEPTH - 1) {
            if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            acc += *(int*)(lds + ((it + 1) % DEPTH) * 32768 + tid * 4);
        }
    }
    if (acc == 0x12345) sink[0] = acc;
}

template <int DEPTH>
void run(const char* a, const char* w, int share, long long a_region, int* sink) {
    const int iters = 192;
    hipFuncSetAttribute((const void*)fill<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, DEPTH * 32768);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    fill<DEPTH><<<256, 512, DEPTH * 32768>>>(a, w, iters, share, a_region, sink);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) fill<DEPTH><<<256, 512, DEPTH * 32768>>>(a, w, iters, share, a_region, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("depth %d (in flight %d x 32 KiB) share %2d: %.3f us per K-step, %.1f GB/s per CU\n", DEPTH, DEPTH - 1, share, ms * 1e3 / iters,
           iters * 32768.0 / ms / 1e6);
}

int main() {
    const long long a_region = 1024LL << 20;
    char *a, *w; hipMalloc(&a, a_region + (64 << 20)); hipMemset(a, 1, a_region + (64 << 20)); hipMalloc(&w, 4 << 20); hipMemset(w, 1, 4 << 20);
    int* sink; hipMalloc(&sink, 4);
    for (int share : {1, 3, 12, 32}) {
        run<2>(a, w, share, a_region, sink); run<3>(a, w, share, a_region, sink); run<4>(a, w, share, a_region, sink); run<5>(a, w, share, a_region, sink);
    }
    return 0;
}
