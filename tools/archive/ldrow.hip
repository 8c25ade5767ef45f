// Micro-benchmark: LDS-DMA feed rate of a GEMM's operand tiles with the REAL access pattern -- rows of RB bytes (one K-step's
// slice of a row) at the matrix row stride -- instead of contiguous kilobytes (tools/ldlat.hip).  RB = 64: an instruction covers
// 16 rows x 64 B (HALF a 128-byte line per row: the other half is requested by the next K-step); RB = 128: 8 rows x 128 B (full
// lines).  One workgroup per CU streams 32 KiB chunks (16 KiB of an activation panel shared by `share` workgroups of its XCD +
// 16 KiB of a weight panel that everybody reads) into a ring of four, three in flight; no compute.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ldrow.bin tools/ldrow.hip && tools/ldrow.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int RB>
__global__ __launch_bounds__(512) void feed(const char* __restrict__ a, const char* __restrict__ w, long long ld, int nt, int ntiles, int share,
                                            int panels_per_round, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
    constexpr int RPI = 1024 / RB;                 // rows per instruction
    constexpr int LPR = RB / 16;                   // lanes per row
    const int lrow = lane / LPR, lpiece = lane % LPR;
    int acc = 0, g = 0;
    for (int t = 0; t < ntiles; ++t) {
        const long long panel = (long long)t * panels_per_round + (xcd * 32 + l) / share;
        const char* ap = a + panel * 256 * ld;
        const char* wp = w + (long long)((l % share) % 3) * 256 * ld;
        const int chunks = RB == 64 ? nt : 2 * (nt / 2);       // RB = 128: two 32 KiB chunks (row halves) per 128-byte column step
        for (int it = 0; it < chunks; ++it, ++g) {
            char* st = lds + (g & 3) * 32768;
            const int col = RB == 64 ? it * 64 : (it >> 1) * 128;
            const int half = RB == 64 ? 0 : (it & 1) * 128;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int rg = wave * 2 + j;
                const int row = RB == 64 ? rg * 16 + lrow : half + rg * 8 + lrow;
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(ap + (long long)row * ld + col + lpiece * 16),
                                                 (void __attribute__((address_space(3)))*)(st + rg * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(wp + (long long)row * ld + col + lpiece * 16),
                                                 (void __attribute__((address_space(3)))*)(st + 16384 + rg * 1024), 16, 0, 0);
            }
            if (g >= 3) {
                asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                acc += *(int*)(lds + ((g + 1) & 3) * 32768 + tid * 4);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x12345) sink[0] = acc;
}

// RB = 64 pieces, but the two halves of every 128-byte line are requested back to back: the pieces of K-steps (2i, 2i+1) are issued
// together, every second iteration (into two ring stages), i.e. the same 32 KiB stages and K-step granularity for a consumer.
__global__ __launch_bounds__(512) void feed_paired(const char* __restrict__ a, const char* __restrict__ w, long long ld, int nt, int ntiles, int share,
                                                   int panels_per_round, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
    const int lrow = lane >> 2, lpiece = lane & 3;
    int acc = 0, g = 0;
    for (int t = 0; t < ntiles; ++t) {
        const long long panel = (long long)t * panels_per_round + (xcd * 32 + l) / share;
        const char* ap = a + panel * 256 * ld;
        const char* wp = w + (long long)((l % share) % 3) * 256 * ld;
        for (int it = 0; it < nt; it += 2, g += 2) {
            char* st0 = lds + (g & 3) * 32768;
            char* st1 = lds + ((g + 1) & 3) * 32768;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int rg = wave * 2 + j;
                const int row = rg * 16 + lrow;
                const char* ga = ap + (long long)row * ld + it * 64 + lpiece * 16;
                const char* gw = wp + (long long)row * ld + it * 64 + lpiece * 16;
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(ga), (void __attribute__((address_space(3)))*)(st0 + rg * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(ga + 64), (void __attribute__((address_space(3)))*)(st1 + rg * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(gw), (void __attribute__((address_space(3)))*)(st0 + 16384 + rg * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(gw + 64), (void __attribute__((address_space(3)))*)(st1 + 16384 + rg * 1024), 16, 0, 0);
            }
            if (g >= 2) {
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // the previous pair has landed, this one flies
                __builtin_amdgcn_s_barrier();
                acc += *(int*)(lds + ((g + 2) & 3) * 32768 + tid * 4);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x12345) sink[0] = acc;
}

void run_paired(const char* a, const char* w, long long ld, int share, int* sink) {
    const int nt = (int)(ld / 64), ntiles = 6, ppr = 256 / share + 1;
    hipFuncSetAttribute((const void*)feed_paired, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    feed_paired<<<256, 512, 4 * 32768>>>(a, w, ld, nt, ntiles, share, ppr, sink);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) feed_paired<<<256, 512, 4 * 32768>>>(a, w, ld, nt, ntiles, share, ppr, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double chunks = (double)ntiles * nt;
    printf("64-byte pairs  K = %4lld  share %2d: %.3f us per 32 KiB, %.1f GB/s per CU, %.2f TB/s chip (%s)\n", ld / 2, share, ms * 1e3 / chunks,
           chunks * 32768.0 / ms / 1e6, chunks * 32768.0 * 256 / ms / 1e9, hipGetErrorString(hipGetLastError()));
}

template <int RB>
void run(const char* a, const char* w, long long ld, int share, int* sink) {
    const int nt = (int)(ld / 64), ntiles = 6, ppr = 256 / share + 1;
    hipFuncSetAttribute((const void*)feed<RB>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    feed<RB><<<256, 512, 4 * 32768>>>(a, w, ld, nt, ntiles, share, ppr, sink);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) feed<RB><<<256, 512, 4 * 32768>>>(a, w, ld, nt, ntiles, share, ppr, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double chunks = (double)ntiles * nt;
    printf("row bytes %3d  K = %4lld  share %2d: %.3f us per 32 KiB, %.1f GB/s per CU, %.2f TB/s chip (%s)\n", RB, ld / 2, share, ms * 1e3 / chunks,
           chunks * 32768.0 / ms / 1e6, chunks * 32768.0 * 256 / ms / 1e9, hipGetErrorString(hipGetLastError()));
}

int main() {
    const long long ldmax = 6144;
    const long long a_bytes = (long long)(6 * 257 + 2) * 256 * ldmax;
    char *a, *w; hipMalloc(&a, a_bytes); hipMemset(a, 1, a_bytes); hipMalloc(&w, 768 * ldmax); hipMemset(w, 1, 768 * ldmax);
    int* sink; hipMalloc(&sink, 4);
    for (long long ld : {1536LL, 6144LL})
        for (int share : {1, 3, 9, 12}) { run<64>(a, w, ld, share, sink); run_paired(a, w, ld, share, sink); run<128>(a, w, ld, share, sink); }
    return 0;
}
