// tools/ntd_probe.hip -- round-6 experiment bench: TWO co-resident four-wave workgroups per CU (csrc/gemm_nt_duo.hip, 256 x 128 tiles, 128 x 64
// wave tiles, de-phased by half a tile) against the one-workgroup-per-CU four-wave kernel of round 5 (csrc/gemm_nt_w128.hip, 256 x 256 tiles),
// at the ViT-B/16 batch-256 shapes (M = 50,432).  Not part of libvitk.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DNTW_PROBE -DNTD_PROBE tools/ntd_probe.hip -o tools/ntd_probe.bin -Lvit_pytorch_amd -lvitk -Wl,-rpath,'$ORIGIN/../vit_pytorch_amd'
//   tools/ntd_probe.bin [rounds]          (run from the repo root)
//
//   * parity: both kernels alone on ALL full 256-row tiles must be BIT-IDENTICAL, per epilogue (C, the second output, the column-sum partial rows);
//   * co-residency: per-workgroup stamps (HW_REG_HW_ID, XCC id, start / end time): how many CUs hold two workgroups, and with opposite phases;
//   * A/B per (shape, epilogue): four-wave kernel | co-resident kernel in phase | with the late start (sweep of the delay);
//   * the ablation table of round 5 on the co-resident kernel: main loop alone / LDS-DMA only / MFMA only / reads only / epilogue only, random
//     and ZERO operands.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <vector>

#define gemm_ntw_serves probe_ntw_serves
#define gemm_ntw_split probe_ntw_split
#define gemm_ntw_launch probe_ntw_launch
#include "../vit_pytorch_amd/csrc/gemm_nt_w128.hip"
#define gemm_ntd_serves probe_ntd_serves
#define gemm_ntd_launch probe_ntd_launch
#include "gemm_nt_duo.hip"

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e__), __FILE__, __LINE__); exit(1); } } while (0)
#define VK(x) do { int r__ = (x); if (r__ != 0) { printf("vitk error %d (%s) at %s:%d\n", r__, vitk_last_error(), __FILE__, __LINE__); exit(1); } } while (0)

static uint16_t f2bf_host(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float frand(uint32_t& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }

struct Buf { void* p = nullptr; size_t n = 0; };
static Buf dalloc(size_t bytes) { Buf b; b.n = bytes; CK(hipMalloc(&b.p, bytes)); return b; }
static Buf rand_bf16(size_t elems, float scale, uint32_t seed) {
    std::vector<uint16_t> h(elems);
    uint32_t s = seed;
    for (size_t i = 0; i < elems; ++i) h[i] = f2bf_host(frand(s) * 2.f * scale);
    Buf b = dalloc(elems * 2);
    CK(hipMemcpy(b.p, h.data(), elems * 2, hipMemcpyHostToDevice));
    return b;
}
static Buf rand_u8(size_t elems, uint32_t seed) {
    std::vector<uint8_t> h(elems);
    uint32_t s = seed;
    for (size_t i = 0; i < elems; ++i) { s = s * 1664525u + 1013904223u; h[i] = (uint8_t)(1 + (s >> 24) % 253); }
    Buf b = dalloc(elems);
    CK(hipMemcpy(b.p, h.data(), elems, hipMemcpyHostToDevice));
    return b;
}

template <typename F> static float time_ms(F&& fn, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) fn();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms / iters;
}
static float median(std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
static size_t diff_bytes(const void* d0, const void* d1, size_t n) {
    std::vector<unsigned char> h0(n), h1(n);
    CK(hipMemcpy(h0.data(), d0, n, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), d1, n, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < n; ++i) bad += h0[i] != h1[i];
    return bad;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 3;
    const bool quick = argc > 2 && atoi(argv[2]) == 1;
    const int64_t M = 50432;
    int dev = 0, cus = 0;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int grid1 = cus / 8 * 8, grid2 = 2 * grid1;
    printf("ntd_probe: M = %lld, %d CUs, grids %d / %d, rounds %d\n", (long long)M, cus, grid1, grid2, rounds);

    struct Shape { const char* name; int64_t N, K; std::vector<int> epis; };
    const std::vector<Shape> shapes = {
        {"FF1   (N 3072, K  768)", 3072, 768, {VITK_EPI_NONE, VITK_EPI_BIAS_GELU_DG8, VITK_EPI_MUL_AUX8, VITK_EPI_BIAS_GELU_DG, VITK_EPI_MUL_AUX}},
        {"FF2   (N  768, K 3072)", 768, 3072, {VITK_EPI_NONE, VITK_EPI_RESID16, VITK_EPI_RESID}},
        {"QKV   (N 2304, K  768)", 2304, 768, {VITK_EPI_NONE}},
        {"out   (N  768, K  768)", 768, 768, {VITK_EPI_RESID16}},
        {"dXqkv (N  768, K 2304)", 768, 2304, {VITK_EPI_NONE}},
    };
    const char* epi_name[10] = {"NONE", "BIAS", "BIAS_GELU", "RESID(f32)", "GELU_BWD", "RESID16", "BIAS_GELU_DG", "MUL_AUX", "BIAS_GELU_DG8", "MUL_AUX8"};
    Buf stamps = dalloc((size_t)grid2 * 4 * 8);
    for (const Shape& sh : shapes) {
        const int64_t N = sh.N, K = sh.K;
        Buf A = rand_bf16(M * K, 1.0f, 1), W = rand_bf16(N * K, 0.05f, 2), bias = rand_bf16(N, 0.5f, 3);
        Buf aux_in = rand_bf16(M * N, 1.0f, 4), aux8_in = rand_u8(M * N, 6), r16 = rand_bf16(M * N, 1.0f, 5);
        Buf r32 = dalloc(M * N * 4);
        CK(hipMemset(r32.p, 0, M * N * 4));
        Buf Wp = dalloc(vitk_pack_w_nt_bytes(N, K));
        VK(vitk_pack_w_nt(W.p, K, N, K, Wp.p, nullptr, nullptr));
        Buf C0 = dalloc(M * N * 4), C1 = dalloc(M * N * 4), X0 = dalloc(M * N * 2), X1 = dalloc(M * N * 2);
        const int64_t R = 2 * ((M + 127) / 128) + 8;
        Buf cs0 = dalloc(R * N * 4), cs1 = dalloc(R * N * 4);
        const int tm_all = (int)(M / 256);
        const double flop = 2.0 * 256 * tm_all * N * K;
        printf("\n== %s: %d full m-tiles; four-wave kernel %.2f rounds of %d tiles, co-resident kernel %.2f rounds of %d ==\n", sh.name, tm_all,
               (double)tm_all * (N / 256) / grid1, grid1, (double)tm_all * (N / 128) / grid2, grid2);
        for (int epi : sh.epis) {
            const bool f32out = epi == VITK_EPI_RESID;
            const bool aux8 = epi == VITK_EPI_BIAS_GELU_DG8 || epi == VITK_EPI_MUL_AUX8;
            const bool aux_is_in = epi == VITK_EPI_MUL_AUX || epi == VITK_EPI_MUL_AUX8;
            const bool aux_is_out = epi == VITK_EPI_BIAS_GELU || epi == VITK_EPI_BIAS_GELU_DG || epi == VITK_EPI_BIAS_GELU_DG8;
            const bool has_bias = epi == VITK_EPI_BIAS || aux_is_out || epi == VITK_EPI_RESID16 || epi == VITK_EPI_RESID;
            const void* resid = epi == VITK_EPI_RESID ? r32.p : (epi == VITK_EPI_RESID16 ? r16.p : nullptr);
            const size_t cbytes = (size_t)256 * tm_all * N * (f32out ? 4 : 2);
            auto aux_of = [&](Buf& X) -> void* { return aux_is_in ? (aux8 ? aux8_in.p : aux_in.p) : (aux_is_out ? X.p : nullptr); };
            auto one = [&](int abl, int dbg, const void* a, const void* w) {      // the four-wave launch alone, all full tiles
                VK(probe_ntw_launch(tm_all, grid1, a, K, w, 0, C0.p, N, N, K, epi, has_bias ? bias.p : nullptr, resid, aux_of(X0),
                                    aux_is_in ? (float*)cs0.p : nullptr, abl, dbg, nullptr));
            };
            auto duo = [&](int abl, int dbg, int delay, void* st, const void* a, const void* w) {      // the co-resident launch alone, all full tiles
                VK(probe_ntd_launch(tm_all, grid2, a, K, w, 0, C1.p, N, N, K, epi, has_bias ? bias.p : nullptr, resid, aux_of(X1),
                                    aux_is_in ? (float*)cs1.p : nullptr, abl, dbg, delay, st, nullptr));
            };
            // ---- parity ----
            CK(hipMemset(C0.p, 0xff, cbytes)); CK(hipMemset(C1.p, 0xee, cbytes));
            CK(hipMemset(X0.p, 0xff, M * N * 2)); CK(hipMemset(X1.p, 0xee, M * N * 2));
            CK(hipMemset(cs0.p, 0, cs0.n)); CK(hipMemset(cs1.p, 0, cs1.n));
            CK(hipMemset(stamps.p, 0, stamps.n));
            one(0, 0, A.p, Wp.p);
            duo(0, 0, -1, stamps.p, A.p, Wp.p);
            CK(hipDeviceSynchronize());
            const size_t bad = diff_bytes(C0.p, C1.p, cbytes);
            const size_t badx = aux_is_out ? diff_bytes(X0.p, X1.p, (size_t)256 * tm_all * N * (aux8 ? 1 : 2)) : 0;
            const size_t badc = aux_is_in ? diff_bytes(cs0.p, cs1.p, (size_t)2 * tm_all * N * 4) : 0;
            printf("  %-13s parity vs the four-wave kernel: C %s (%zu bytes differ)", epi_name[epi], bad ? "MISMATCH" : "bit-identical", bad);
            if (aux_is_out) printf(", second output %s (%zu)", badx ? "MISMATCH" : "bit-identical", badx);
            if (aux_is_in) printf(", column-sum rows %s (%zu)", badc ? "MISMATCH" : "bit-identical", badc);
            printf("\n");
            // the same through the row-major W path (ldw != 0)
            if (epi == sh.epis[0]) {
                CK(hipMemset(C1.p, 0xee, cbytes));
                VK(probe_ntd_launch(tm_all, grid2, A.p, K, W.p, K, C1.p, N, N, K, epi, has_bias ? bias.p : nullptr, resid, aux_of(X1),
                                    aux_is_in ? (float*)cs1.p : nullptr, 0, 0, -1, nullptr, nullptr));
                CK(hipDeviceSynchronize());
                printf("  %-13s row-major W: C %s\n", epi_name[epi], diff_bytes(C0.p, C1.p, cbytes) ? "MISMATCH" : "bit-identical");
            }
            // ---- co-residency from the stamps of that launch ----
            if (epi == sh.epis[0]) {
                std::vector<unsigned long long> st((size_t)grid2 * 4);
                CK(hipMemcpy(st.data(), stamps.p, st.size() * 8, hipMemcpyDeviceToHost));
                std::map<unsigned long long, std::vector<int>> per_cu;
                unsigned long long t0 = ~0ULL, t1 = 0;
                for (int b = 0; b < grid2; ++b) {
                    const unsigned hw = (unsigned)st[4 * b], xcc = (unsigned)(st[4 * b] >> 32);
                    const unsigned long long key = ((unsigned long long)(xcc & 0xf) << 32) | (hw & 0xff00u);   // xcc, cu_id[11:8], sh_id[12], se_id[15:13]
                    per_cu[key].push_back(b);
                    if (st[4 * b + 2]) t0 = std::min(t0, st[4 * b + 2]);
                    t1 = std::max(t1, st[4 * b + 3]);
                }
                int cu2 = 0, cu_opp = 0, cu_other = 0, late_first_half = 0, late_second_half = 0;
                for (auto& kv : per_cu) {
                    if (kv.second.size() == 2) { ++cu2; if (st[4 * kv.second[0] + 1] != st[4 * kv.second[1] + 1]) ++cu_opp; }
                    else ++cu_other;
                }
                for (int b = 0; b < grid2; ++b) { if (st[4 * b + 1]) { if (b < grid2 / 2) ++late_first_half; else ++late_second_half; } }
                printf("  co-residency: %zu distinct (xcc, se, cu) ids; %d hold exactly two workgroups, %d of those with opposite phases; %d hold another count;\n"
                       "                late workgroups: %d in the first half of the grid, %d in the second; launch span %.1f us (100 MHz ticks)\n",
                       per_cu.size(), cu2, cu_opp, cu_other, late_first_half, late_second_half, (double)(t1 - t0) / 100.0);
                // a sample: hw ids of the first 4 workgroups and of their grid / 2 partners
                for (int b = 0; b < 4; ++b)
                    printf("                block %3d hw_id %08x xcc %u late %llu | block %3d hw_id %08x xcc %u late %llu\n", b, (unsigned)st[4 * b], (unsigned)(st[4 * b] >> 32),
                           st[4 * b + 1], b + grid2 / 2, (unsigned)st[4 * (b + grid2 / 2)], (unsigned)(st[4 * (b + grid2 / 2)] >> 32), st[4 * (b + grid2 / 2) + 1]);
            }
            // ---- A/B timing, interleaved rounds ----
            struct Var { const char* name; int dbg, delay; };
            std::vector<Var> vars = {{"in phase (no late start)", 4, 0}, {"late start: default delay", 0, -1}, {"late start by grid half, default delay", 2, -1}};
            if (!quick) for (int d : {4, 8, 12, 16, 24, 32, 48}) vars.push_back({"late start, delay (x ~1024 cycles)", 0, d});
            std::vector<std::vector<float>> t(vars.size() + 1);
            for (int r = 0; r <= rounds; ++r) {
                const float a = time_ms([&] { one(0, 0, A.p, Wp.p); }, 10);
                if (r) t[0].push_back(a);
                for (size_t v = 0; v < vars.size(); ++v) {
                    const float b = time_ms([&] { duo(0, vars[v].dbg, vars[v].delay, nullptr, A.p, Wp.p); }, 10);
                    if (r) t[v + 1].push_back(b);
                }
            }
            const float m1 = median(t[0]);
            printf("  %-13s four-wave kernel, one workgroup per CU          %7.1f us %7.1f TF/s\n", epi_name[epi], m1 * 1e3, flop / m1 / 1e9);
            for (size_t v = 0; v < vars.size(); ++v) {
                const float m2 = median(t[v + 1]);
                if (vars[v].delay > 0) printf("  %-13s two per CU, %-34s %3d %7.1f us %7.1f TF/s (x%.3f)\n", epi_name[epi], vars[v].name, vars[v].delay, m2 * 1e3, flop / m2 / 1e9, m1 / m2);
                else printf("  %-13s two per CU, %-38s %7.1f us %7.1f TF/s (x%.3f)\n", epi_name[epi], vars[v].name, m2 * 1e3, flop / m2 / 1e9, m1 / m2);
            }
            if (quick) continue;
            // ---- ablations (timing only) ----
            struct Ab { const char* name; int abl, dbg; };
            if (epi == VITK_EPI_NONE) {
                const Ab abs[] = {{"main loop alone (no epilogue)", 0, 1}, {"LDS-DMA only", 6, 1}, {"MFMA only", 3, 1}, {"fragment reads only", 5, 1},
                                  {"DMA + MFMA", 2, 1}, {"reads + MFMA", 1, 1}, {"empty loop", 7, 1}, {"epilogue only (empty loop + stores)", 7, 0}};
                Buf Az = dalloc(M * K * 2), Wz = dalloc(vitk_pack_w_nt_bytes(N, K));
                CK(hipMemset(Az.p, 0, Az.n)); CK(hipMemset(Wz.p, 0, Wz.n));
                printf("      %-38s %21s | %21s\n", "", "four-wave: random   zero", "two per CU: random   zero");
                for (const Ab& ab : abs) {
                    std::vector<float> a, az, b, bz;
                    for (int r = 0; r < rounds; ++r) {
                        a.push_back(time_ms([&] { one(ab.abl, ab.dbg, A.p, Wp.p); }, 10));
                        az.push_back(time_ms([&] { one(ab.abl, ab.dbg, Az.p, Wz.p); }, 10));
                        b.push_back(time_ms([&] { duo(ab.abl, ab.dbg, -1, nullptr, A.p, Wp.p); }, 10));
                        bz.push_back(time_ms([&] { duo(ab.abl, ab.dbg, -1, nullptr, Az.p, Wz.p); }, 10));
                    }
                    printf("      %-38s %10.1f %10.1f | %10.1f %10.1f us\n", ab.name, median(a) * 1e3, median(az) * 1e3, median(b) * 1e3, median(bz) * 1e3);
                }
                CK(hipFree(Az.p)); CK(hipFree(Wz.p));
            } else {
                std::vector<float> a, b, c;
                for (int r = 0; r < rounds; ++r) {
                    a.push_back(time_ms([&] { one(7, 0, A.p, Wp.p); }, 10));
                    b.push_back(time_ms([&] { duo(7, 0, -1, nullptr, A.p, Wp.p); }, 10));
                    c.push_back(time_ms([&] { duo(7, 4, 0, nullptr, A.p, Wp.p); }, 10));
                }
                printf("      epilogue only (empty loop + epilogue): four-wave %7.1f us | two per CU %7.1f us (in phase %7.1f)\n", median(a) * 1e3, median(b) * 1e3, median(c) * 1e3);
            }
        }
        for (Buf* b : {&A, &W, &bias, &aux_in, &aux8_in, &r16, &r32, &Wp, &C0, &C1, &X0, &X1, &cs0, &cs1}) CK(hipFree(b->p));
    }
    return 0;
}
