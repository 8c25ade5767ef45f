// tools/stbw.hip -- what does a CU's global-store path deliver?  (round 4: the NT GEMM epilogues are priced by it)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/stbw.hip -o tools/stbw.bin && tools/stbw.bin
// One 512-thread workgroup per CU (8 waves, like the NT GEMM), each wave storing full 128-byte lines (8 lines per dwordx4 instruction) to its
// own contiguous region; variants: instruction flavour (plain / nt / sc1 / sc0 sc1 / buffer_store), waves that store (8, 4, 2, 1), and the
// number of CUs that store at all (256, 128, 64, 32, 8 -- is ~14 B/clk a per-CU limit or the chip's write bandwidth divided by 256?),
// plus the same with a concurrent LDS-DMA read stream from L2-resident data (do loads and stores share the path?).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e__), __LINE__); return 1; } } while (0)

template <int FLAVOUR>
__device__ __forceinline__ void st16(char* p, f32x4 v) {
    if constexpr (FLAVOUR == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    else if constexpr (FLAVOUR == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
    else if constexpr (FLAVOUR == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else if constexpr (FLAVOUR == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
}

// each storing wave writes `kib` KiB (one KiB per instruction), 16 instructions between waits
template <int FLAVOUR>
__global__ __launch_bounds__(512) void store_kernel(char* out, long long bytes_per_wg, int kib, int waves, int active_wgs, const char* rd, int rd_kib) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if ((int)blockIdx.x >= active_wgs && rd_kib == 0) return;
    const bool stores = (int)blockIdx.x < active_wgs && wave < waves;
    char* p = out + (long long)blockIdx.x * bytes_per_wg + (long long)wave * kib * 1024 + lane * 16;
    f32x4 v = {(float)lane, 1.f, 2.f, 3.f};
    // optional LDS-DMA read stream: every wave reads rd_kib KiB (1 KiB per instruction) from a 2 MiB region every workgroup shares
    const char* rp = rd + ((wave * 64 + lane) * 16) % (2 << 20);
    int rleft = rd_kib;
    for (int i = 0; i < kib || rleft > 0; ++i) {
        if (stores && i < kib) { st16<FLAVOUR>(p, v); p += 1024; }
        if (rleft > 0) {
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(rp + ((long long)i * 8192) % (2 << 20)),
                                             (void __attribute__((address_space(3)))*)(lds + wave * 4096 + (i & 3) * 1024), 16, 0, 0);
            --rleft;
        }
        if ((i & 15) == 15) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// the NT GEMM epilogue's pattern: a workgroup writes 256 x 256 tiles of 16-bit elements into a row-major matrix with `ld_bytes` per row:
// wave (wm, wn) owns rows wm * 128 .. + 127 and the 128-byte piece wn of every 512-byte row segment; one instruction = 8 rows x one line.
// tiles_per_wg tiles per workgroup, consecutive tiles ld-adjacent in n (like the grouped order) -- vs the same bytes written contiguously.
__global__ __launch_bounds__(512) void tile_store_kernel(char* out, long long ld_bytes, int tiles_per_wg, int tiles_n, int contiguous) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave >> 2, wn = wave & 3;
    f32x4 v = {(float)lane, 1.f, 2.f, 3.f};
    for (int t = 0; t < tiles_per_wg; ++t) {
        const long long tile = (long long)blockIdx.x * tiles_per_wg + t;
        if (contiguous) {
            char* p = out + tile * 131072 + wave * 16384 + lane * 16;
#pragma unroll 4
            for (int i = 0; i < 16; ++i) st16<0>(p + i * 1024, v);
        } else {
            const long long tm = tile / tiles_n, tn = tile % tiles_n;
            char* p = out + (tm * 256 + wm * 128 + (lane >> 3)) * ld_bytes + tn * 512 + wn * 128 + (lane & 7) * 16;
#pragma unroll 4
            for (int i = 0; i < 16; ++i) st16<0>(p + (long long)i * 8 * ld_bytes, v);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

typedef void (*kern_t)(char*, long long, int, int, int, const char*, int);

int main() {
    const int kib = 2048;                                   // per wave: 2 MiB; per workgroup up to 16 MiB
    const long long per_wg = 8LL * kib * 1024;
    char* out; CK(hipMalloc(&out, per_wg * 256));
    char* rd; CK(hipMalloc(&rd, 4 << 20)); CK(hipMemset(rd, 1, 4 << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    kern_t kerns[5] = {store_kernel<0>, store_kernel<1>, store_kernel<2>, store_kernel<3>, store_kernel<4>};
    const char* fl[5] = {"plain", "nt", "sc1", "sc0 sc1", "sc0"};
    for (int a = 0; a < 5; ++a) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kerns[a]), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    auto run = [&](int f, int waves, int wgs, int rd_kib, int st_kib) -> float {
        float ms = 0, best = 1e9;
        for (int r = 0; r < 3; ++r) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kerns[f], dim3(256), dim3(512), 65536, 0, out, per_wg, st_kib, waves, wgs, rd, rd_kib);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        return best;
    };
    printf("flavours, 8 waves per CU store, 256 CUs (2 MiB per wave):\n");
    for (int f = 0; f < 5; ++f) {
        const float ms = run(f, 8, 256, 0, kib);
        const double bytes = 256.0 * 8 * kib * 1024;
        printf("  %-8s %.3f ms  %.2f TB/s  %.1f GB/s per CU\n", fl[f], ms, bytes / ms / 1e9, bytes / 256 / ms / 1e6);
    }
    printf("storing waves per CU (plain, 256 CUs, 2 MiB per storing wave):\n");
    for (int w : {8, 4, 2, 1}) {
        const float ms = run(0, w, 256, 0, kib);
        const double bytes = 256.0 * w * kib * 1024;
        printf("  %d waves  %.3f ms  %.2f TB/s  %.1f GB/s per CU\n", w, ms, bytes / ms / 1e9, bytes / 256 / ms / 1e6);
    }
    printf("CUs that store (plain, 8 waves):\n");
    for (int c : {256, 128, 64, 32, 8}) {
        const float ms = run(0, 8, c, 0, kib);
        const double bytes = (double)c * 8 * kib * 1024;
        printf("  %3d CUs  %.3f ms  %.2f TB/s  %.1f GB/s per storing CU\n", c, ms, bytes / ms / 1e9, bytes / c / ms / 1e6);
    }
    printf("LDS-DMA reads of L2-resident data alone / stores alone / both in one loop (8 waves, 256 CUs, 1 MiB each way per wave):\n");
    {
        const float r = run(0, 0, 256, 1024, 1024), s = run(0, 8, 256, 0, 1024), b = run(0, 8, 256, 1024, 1024);
        const double bytes = 256.0 * 8 * 1024 * 1024;
        printf("  reads %.3f ms (%.1f GB/s per CU)  stores %.3f ms (%.1f GB/s per CU)  both %.3f ms (sum of the two alone: %.3f)\n",
               r, bytes / 256 / r / 1e6, s, bytes / 256 / s / 1e6, b, r + s);
    }
    printf("the NT epilogue's store pattern (256 CUs, 9 tiles of 256 x 256 x 2 B per workgroup = 302 MB), by row stride:\n");
    {
        char* big; CK(hipMalloc(&big, 50432LL * 6144 + (1 << 20)));
        for (int mode = 0; mode < 4; ++mode) {
            const long long ld = mode == 1 ? 1536 : mode == 2 ? 4608 : 6144;       // N = 768 / 2304 / 3072 columns of 2 bytes
            const int tiles_n = (int)(ld / 512), contiguous = mode == 0;
            float ms = 0, best = 1e9;
            for (int r = 0; r < 3; ++r) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(tile_store_kernel, dim3(256), dim3(512), 0, 0, big, ld, 9, tiles_n, contiguous);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double bytes = 256.0 * 9 * 131072;
            printf("  %-28s %.3f ms  %.2f TB/s  %.1f GB/s per CU\n", contiguous ? "contiguous tiles" : mode == 1 ? "row stride 1536 B (N = 768)" : mode == 2 ? "row stride 4608 B (N = 2304)" : "row stride 6144 B (N = 3072)",
                   best, bytes / best / 1e9, bytes / 256 / best / 1e6);
        }
        // one workgroup per XCD-quarter: 64 CUs only (is the strided pattern slow per CU or only in aggregate?)
        for (int mode = 0; mode < 2; ++mode) {
            float ms = 0, best = 1e9;
            for (int r = 0; r < 3; ++r) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(tile_store_kernel, dim3(64), dim3(512), 0, 0, big, 6144LL, 9, 12, mode == 0);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double bytes = 64.0 * 9 * 131072;
            printf("  64 CUs, %-20s %.3f ms  %.2f TB/s  %.1f GB/s per CU\n", mode == 0 ? "contiguous" : "row stride 6144 B", best, bytes / best / 1e9, bytes / 64 / best / 1e6);
        }
    }
    return 0;
}
