// Micro-benchmark (follow-up of tools/ldrow.hip): LDS-DMA feed of one workgroup per CU with the weight half of every 32 KiB K-step
// K-BLOCKED (one contiguous KiB per instruction, as vitk_pack_w_nt lays it out) and the activation half as 64-byte row pieces,
// plus two ways of helping the activation stream:
//   mode 0  baseline (what the persistent NT kernel issues today)
//   mode 1  + L2 "touches": a 4-byte LDS-DMA per 128-byte line of the activation rows, TOUCH_AHEAD K-steps ahead of the DMA, by every
//             workgroup
//   mode 2  + the same touches, issued by ONE workgroup of the `share` that read the same panel (rotating every 8 K-steps)
//   mode 3  activation rows as 128-byte pieces (8 rows x 128 B per instruction), a pair of K-steps issued every second step
//   hipcc --offload-arch=gfx950 -O3 -o tools/ldrow2.bin tools/ldrow2.hip && tools/ldrow2.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int TOUCH_AHEAD = 8;

template <int MODE>
__global__ __launch_bounds__(512) void feed(const char* __restrict__ a, const char* __restrict__ w, long long ld, int nt, int ntiles, int share,
                                            int panels_per_round, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
    const int lrow = lane >> 2, lpiece = lane & 3;
    char* dump = lds + 4 * 32768;
    int acc = 0, g = 0;
    for (int t = 0; t < ntiles; ++t) {
        const long long panel = (long long)t * panels_per_round + (xcd * 32 + l) / share;
        const char* ap = a + panel * 256 * ld;
        const char* wp = w + (long long)((l % share) % 3) * nt * 16384;          // K-blocked: (tile, kt) -> 16 KiB
        for (int it = 0; it < nt; ++it, ++g) {
            char* st = lds + (g & 3) * 32768;
            if (MODE == 3) {
                if ((it & 1) == 0) {
                    // pair (it, it + 1): 32 KiB of A = 32 instructions of 8 rows x 128 B, 4 per wave; LDS block = [8 rows x 64 B of it][.. of it + 1]
                    char* pa = lds + ((g >> 1) & 1) * 65536;      // two pair regions of 32 KiB (A) inside the 128 KiB (W keeps its own 4 x 16 KiB)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int blk = wave * 4 + j;             // 32 blocks of 8 rows
                        const int half = lane >> 5, r8 = (lane >> 2) & 7;
                        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(ap + (long long)(blk * 8 + r8) * ld + it * 64 + half * 64 + lpiece * 16),
                                                         (void __attribute__((address_space(3)))*)(pa + blk * 1024), 16, 0, 0);
                    }
                }
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(wp + (long long)it * 16384 + (wave * 2 + j) * 1024 + lane * 16),
                                                     (void __attribute__((address_space(3)))*)(lds + 131072 - 65536 + (g & 3) * 16384 + (wave * 2 + j) * 1024), 16, 0, 0);
                if (g >= 3) {
                    if (it & 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    acc += *(int*)(lds + tid * 4);
                }
                continue;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int rg = wave * 2 + j;
                const int row = rg * 16 + lrow;
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(ap + (long long)row * ld + it * 64 + lpiece * 16),
                                                 (void __attribute__((address_space(3)))*)(st + rg * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(wp + (long long)it * 16384 + rg * 1024 + lane * 16),
                                                 (void __attribute__((address_space(3)))*)(st + 16384 + rg * 1024), 16, 0, 0);
            }
            if ((MODE == 1 || MODE == 2) && (it & 7) == 0) {
                // lines of K-steps it + AHEAD .. it + AHEAD + 7 of this wave's 32 rows: the 4 lanes of a row take 4 consecutive 128-byte lines
                const bool mine = MODE == 1 || ((it >> 3) % share) == (l % share);
                const int col = (it + TOUCH_AHEAD) * 64 + lpiece * 128;
                if (mine && col < ld) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int row = (wave * 2 + j) * 16 + lrow;
                        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(ap + (long long)row * ld + col),
                                                         (void __attribute__((address_space(3)))*)(dump), 4, 0, 0);
                    }
                }
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

// mode 4: vmcnt retires in order PER WAVE, so a wave that issues both operands can keep no more K-steps of the activation
// stream in flight than of the weight stream.  Here waves 0-3 issue only activation pieces (ring of DA stages, DA - 1 in flight)
// and waves 4-7 only weight pieces (ring of DW stages): the activation stream -- the one that misses L2 -- gets the deep window.
template <int DA, int DW>
__global__ __launch_bounds__(512) void feed_split(const char* __restrict__ a, const char* __restrict__ w, long long ld, int nt, int ntiles, int share,
                                                  int panels_per_round, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
    const int lrow = lane >> 2, lpiece = lane & 3;
    char* wbase = lds + DA * 16384;
    int acc = 0, g = 0;
    for (int t = 0; t < ntiles; ++t) {
        const long long panel = (long long)t * panels_per_round + (xcd * 32 + l) / share;
        const char* ap = a + panel * 256 * ld;
        const char* wp = w + (long long)((l % share) % 3) * nt * 16384;
        for (int it = 0; it < nt; ++it, ++g) {
            if (wave < 4) {
                char* st = lds + (g % DA) * 16384;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int rg = wave * 4 + j;
                    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(ap + (long long)(rg * 16 + lrow) * ld + it * 64 + lpiece * 16),
                                                     (void __attribute__((address_space(3)))*)(st + rg * 1024), 16, 0, 0);
                }
                if (g >= DA - 1) {
                    if (DA == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                    else if (DA == 5) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                    else if (DA == 6) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                }
            } else {
                char* st = wbase + (g % DW) * 16384;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int rg = (wave - 4) * 4 + j;
                    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(wp + (long long)it * 16384 + rg * 1024 + lane * 16),
                                                     (void __attribute__((address_space(3)))*)(st + rg * 1024), 16, 0, 0);
                }
                if (g >= DW - 1) {
                    if (DW == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else if (DW == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                }
            }
            __builtin_amdgcn_s_barrier();
            acc += *(int*)(lds + tid * 4);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x12345) sink[0] = acc;
}

template <int DA, int DW>
void run_split(const char* a, const char* w, long long ld, int share, int* sink) {
    const int nt = (int)(ld / 64), ntiles = 6, ppr = 256 / share + 1;
    const int ldsb = (DA + DW) * 16384;
    (void)hipFuncSetAttribute((const void*)feed_split<DA, DW>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    feed_split<DA, DW><<<256, 512, ldsb>>>(a, w, ld, nt, ntiles, share, ppr, sink);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) feed_split<DA, DW><<<256, 512, ldsb>>>(a, w, ld, nt, ntiles, share, ppr, sink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double chunks = (double)ntiles * nt;
    printf("split waves: A ring %d, W ring %d  K = %4lld  share %2d: %.3f us per K-step, %.1f GB/s per CU (%s)\n", DA, DW, ld / 2, share,
           ms * 1e3 / chunks, chunks * 32768.0 / ms / 1e6, hipGetErrorString(hipGetLastError()));
}

template <int MODE>
void run(const char* a, const char* w, long long ld, int share, int* sink) {
    const int nt = (int)(ld / 64), ntiles = 6, ppr = 256 / share + 1;
    const int ldsb = 4 * 32768 + 1024;
    (void)hipFuncSetAttribute((const void*)feed<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    feed<MODE><<<256, 512, ldsb>>>(a, w, ld, nt, ntiles, share, ppr, sink);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) feed<MODE><<<256, 512, ldsb>>>(a, w, ld, nt, ntiles, share, ppr, sink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double chunks = (double)ntiles * nt;
    printf("mode %d  K = %4lld  share %2d: %.3f us per K-step, %.1f GB/s per CU (%s)\n", MODE, ld / 2, share, ms * 1e3 / chunks,
           chunks * 32768.0 / ms / 1e6, hipGetErrorString(hipGetLastError()));
}

int main() {
    const long long ldmax = 6144;
    const long long a_bytes = (long long)(6 * 257 + 2) * 256 * ldmax;
    char *a, *w; (void)hipMalloc(&a, a_bytes); (void)hipMemset(a, 1, a_bytes); (void)hipMalloc(&w, 768 * ldmax); (void)hipMemset(w, 1, 768 * ldmax);
    int* sink; (void)hipMalloc(&sink, 4);
    for (long long ld : {1536LL, 4608LL, 6144LL})
        for (int share : {3, 9, 12}) {
            run<0>(a, w, ld, share, sink);
            if (getenv("LDROW2_ALL")) { run<1>(a, w, ld, share, sink); run<2>(a, w, ld, share, sink); run<3>(a, w, ld, share, sink); }
            run_split<4, 4>(a, w, ld, share, sink); run_split<5, 3>(a, w, ld, share, sink); run_split<6, 2>(a, w, ld, share, sink);
        }
    return 0;
}
