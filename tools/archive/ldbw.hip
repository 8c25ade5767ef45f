// Micro-benchmark: per-CU fill rate HBM/L2 -> LDS, LDS-DMA (global_load_lds dwordx4) vs register staging
// (global_load_dwordx4 + ds_write_b128).  Each 512-thread workgroup (one per CU) streams `iters` tiles of 32 KiB
// from a region of `region_kb` KiB (small = L2-resident, large = HBM), several tiles in flight.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) int i32x4;

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void fill(const char* __restrict__ src, long long region, int iters, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long base = (MODE == 3) ? 0 : ((long long)blockIdx.x * 4096 * 131) % region;  // different start per WG (MODE 3: the same for all)
    i32x4 r[DEPTH][4];
    int acc = 0;
    for (int it = 0; it < iters; ++it) {
        const long long off = (base + (long long)it * 32768) % region;
        char* st = lds + (it % DEPTH) * 32768;
        if (MODE == 2) {      // GEMM-like: half of the tile is a panel every workgroup reads at the same time, half is private
            const long long shared_off = ((long long)it * 16384) % region;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const char* p = (j < 2) ? src + shared_off + (wave * 2 + j) * 1024 + lane * 16 : src + off + (wave * 2 + j - 2) * 1024 + lane * 16;
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)p,
                                                 (void __attribute__((address_space(3)))*)(st + (wave * 4 + j) * 1024), 16, 0, 0);
            }
            if (it >= DEPTH - 1) {
                asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                acc += *(int*)(lds + ((it + 1) % DEPTH) * 32768 + tid * 4);
            }
        } else if (MODE == 0 || MODE == 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + off + (wave * 4 + j) * 1024 + lane * 16),
                                                 (void __attribute__((address_space(3)))*)(st + (wave * 4 + j) * 1024), 16, 0, 0);
            if (it >= DEPTH - 1) {
                if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                acc += *(int*)(lds + ((it + 1) % DEPTH) * 32768 + tid * 4);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) r[it % DEPTH][j] = *(const i32x4*)(src + off + (wave * 4 + j) * 1024 + lane * 16);
            if (it >= DEPTH - 1) {
                const int o = (it + 1) % DEPTH;
#pragma unroll
                for (int j = 0; j < 4; ++j) *(i32x4*)(lds + o * 32768 + (wave * 4 + j) * 1024 + lane * 16) = r[o][j];
                __syncthreads();
                acc += *(int*)(lds + o * 32768 + tid * 4);
            }
        }
    }
    if (acc == 0x12345) sink[0] = acc;
}

template <int MODE, int DEPTH>
void run(const char* name, const char* d, long long region, int iters, int* sink) {
    hipFuncSetAttribute((const void*)fill<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, DEPTH * 32768);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) fill<MODE, DEPTH><<<256, 512, DEPTH * 32768>>>(d, region, iters, sink);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) fill<MODE, DEPTH><<<256, 512, DEPTH * 32768>>>(d, region, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double bytes = 256.0 * iters * 32768;
    printf("%-28s region %8lld KiB: %.3f ms  %.2f TB/s  (%.1f GB/s per CU)\n", name, region / 1024, ms, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
}

int main() {
    const long long big = 2048LL << 20;
    char* d; hipMalloc(&d, big); hipMemset(d, 1, big);
    int* sink; hipMalloc(&sink, 4);
    for (long long region : {2LL << 20, 64LL << 20, big}) {   // 2 MiB (L2), 64 MiB (MALL), 2 GiB (HBM)
        run<0, 4>("lds-dma depth4", d, region, 512, sink);
        run<3, 4>("lds-dma depth4 SAME addr", d, region, 512, sink);
        run<2, 4>("lds-dma depth4 half shared", d, region, 512, sink);
        run<0, 2>("lds-dma depth2", d, region, 512, sink);
        run<1, 2>("reg-stage depth2", d, region, 512, sink);
        run<1, 4>("reg-stage depth4", d, region, 512, sink);
    }
    return 0;
}
