// tools/nt_probe.hip -- round-5 experiment bench for the NT GEMM  C[M,N] = A[M,K] . W[N,K]^T (+ epilogue)  (not part of libvitk).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DNTW_PROBE tools/nt_probe.hip -o tools/nt_probe.bin -Lvit_pytorch_amd -lvitk -Wl,-rpath,'$ORIGIN/../vit_pytorch_amd'
//   tools/nt_probe.bin [rounds]          (run from the repo root)
//
// Kernel under test: vit_pytorch_amd/csrc/gemm_nt_w128.hip (included here with its ablation instances: the product library builds
// only ABL = 0) -- four waves x 128 x 128 wave tiles, asm-pinned MFMAs, two fragment sets, buffer-descriptor LDS-DMA -- against the
// 8-wave persistent kernel of rounds 2-4 (libvitk.so with VITK_NT_W128=0), at the ViT-B/16 batch-256 shapes (M = 50,432).
//   * parity: the product path (four-wave launch on the whole rounds + 8-wave launch on the remaining rows) must be BIT-IDENTICAL to the
//     8-wave kernel alone, per epilogue;
//   * A/B per (shape, epilogue): 8-wave | product path | four-wave launch alone on ALL full tiles;
//   * ablations of the four-wave main loop (timing only, results are wrong): main loop alone (no epilogue) | LDS-DMA only | MFMA only |
//     fragment reads only | empty loop | epilogue only; strict vs exact-count waits after an epilogue.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#define gemm_ntw_serves probe_ntw_serves
#define gemm_ntw_split probe_ntw_split
#define gemm_ntw_launch probe_ntw_launch
#include "../vit_pytorch_amd/csrc/gemm_nt_w128.hip"
#define NTX_PROBE
#define gemm_ntx_serves probe_ntx_serves
#define gemm_ntx_launch probe_ntx_launch
#include "gemm_nt_x2.hip"

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

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IOLBF, 0);      // (a GPU fault must not take the last lines with it)
    const int rounds = argc > 1 ? atoi(argv[1]) : 3;
    const int64_t M = 50432;
    int dev = 0, cus = 0;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int grid = cus / 8 * 8;
    printf("nt_probe: M = %lld, %d CUs, grid %d, rounds %d\n", (long long)M, cus, grid, rounds);

    if (argc > 2 && atoi(argv[2]) == 2) {
        // mode 2: why is FF1 slower inside the training step than in this bench?  The same launch (a) on ONE pair of output buffers, as
        // everywhere below, (b) rotating over NB pairs (fresh output memory every launch, like the step: 15 GB of saved activations),
        // (c) many launches back to back (sustained load); each with the plain-store epilogue and the FF1 epilogue.
        const int64_t N = 3072, K = 768;
        Buf A = rand_bf16(M * K, 1.0f, 1), W = rand_bf16(N * K, 0.05f, 2), bias = rand_bf16(N, 0.5f, 3);
        Buf Wp = dalloc(vitk_pack_w_nt_bytes(N, K));
        VK(vitk_pack_w_nt(W.p, K, N, K, Wp.p, nullptr, nullptr));
        const int NB = 16;
        std::vector<Buf> Cs, Xs;
        for (int i = 0; i < NB; ++i) { Cs.push_back(dalloc(M * N * 2)); Xs.push_back(dalloc(M * N * 2)); }
        const int epis[3] = {VITK_EPI_NONE, VITK_EPI_BIAS_GELU_DG8, VITK_EPI_BIAS_GELU_DG};
        if (argc > 3) {     // sustained load: seconds of the FF1 launch back to back, 500 launches per line
            for (int blk = 0; blk < atoi(argv[3]); ++blk)
                printf("  sustained FF1 (epilogue 8), launches %5d ..: %7.1f us\n", blk * 500, 1e3 * time_ms([&] {
                    VK(vitk_gemm_nt_bf16(A.p, K, Wp.p, 0, Cs[0].p, N, M, N, K, VITK_EPI_BIAS_GELU_DG8, bias.p, nullptr, Xs[0].p, nullptr)); }, 500));
            return 0;
        }
        for (int epi : epis) {
            for (int nb : {1, NB}) {
                for (int iters : {10, 200}) {
                    int it = 0;
                    std::vector<float> t;
                    for (int r = 0; r < rounds; ++r)
                        t.push_back(time_ms([&] { const int b = (it++) % nb;
                            VK(vitk_gemm_nt_bf16(A.p, K, Wp.p, 0, Cs[b].p, N, M, N, K, epi, epi ? bias.p : nullptr, nullptr, epi ? Xs[b].p : nullptr, nullptr)); }, iters));
                    printf("  epilogue %d  output buffers %2d  launches back to back %3d: %7.1f us\n", epi, nb, iters, median(t) * 1e3);
                }
            }
        }
        return 0;
    }
    struct Shape { const char* name; int64_t N, K; std::vector<int> epis; };
    const std::vector<Shape> shapes = {
        {"FF1   (N 3072, K  768)", 3072, 768, {VITK_EPI_NONE, VITK_EPI_BIAS_GELU_DG, VITK_EPI_BIAS_GELU_DG8, VITK_EPI_MUL_AUX, VITK_EPI_MUL_AUX8, VITK_EPI_BIAS_GELU, VITK_EPI_GELU_BWD}},
        {"FF2   (N  768, K 3072)", 768, 3072, {VITK_EPI_NONE, VITK_EPI_RESID16, VITK_EPI_RESID, VITK_EPI_BIAS}},
        {"QKV   (N 2304, K  768)", 2304, 768, {VITK_EPI_NONE}},
        {"out   (N  768, K  768)", 768, 768, {VITK_EPI_NONE, VITK_EPI_RESID16}},
        {"dXqkv (N  768, K 2304)", 768, 2304, {VITK_EPI_NONE}},
    };
    const char* epi_name[10] = {"NONE", "BIAS", "BIAS_GELU", "RESID(f32)", "GELU_BWD", "RESID16", "BIAS_GELU_DG", "MUL_AUX", "BIAS_GELU_DG8", "MUL_AUX8"};
    const bool quick = argc > 2 && atoi(argv[2]) == 1;      // no ablations
    double sum_old = 0, sum_new = 0;
    for (const Shape& sh : shapes) {
        const int64_t N = sh.N, K = sh.K;
        Buf A = rand_bf16(M * K, 1.0f, 1), W = rand_bf16(N * K, 0.05f, 2), bias = rand_bf16(N, 0.5f, 3);
        Buf aux_in = rand_bf16(M * N, 1.0f, 4), r16 = rand_bf16(M * N, 1.0f, 5);
        Buf r32 = dalloc(M * N * 4);
        CK(hipMemset(r32.p, 0, M * N * 4));
        Buf Wp = dalloc(vitk_pack_w_nt_bytes(N, K));
        VK(vitk_pack_w_nt(W.p, K, N, K, Wp.p, nullptr, nullptr));
        Buf C0 = dalloc(M * N * 4), C1 = dalloc(M * N * 4), X0 = dalloc(M * N * 2), X1 = dalloc(M * N * 2);
        const int64_t R = 2 * ((M + 127) / 128) + 8;
        Buf cs0 = dalloc(R * N * 4), cs1 = dalloc(R * N * 4);
        const double flop = 2.0 * M * N * K;
        const int tm_all = (int)(M / 256), tm_split = probe_ntw_split(M, N, K, grid);
        printf("\n== %s: %d full m-tiles x %lld n-tiles = %.2f rounds; product split: four-wave %d m-tiles, 8-wave %lld rows ==\n", sh.name, tm_all,
               (long long)(N / 256), (double)tm_all * (N / 256) / grid, tm_split, (long long)(M - 256LL * tm_split));
        for (int epi : sh.epis) {
            const bool f32out = epi == VITK_EPI_RESID;
            const bool aux_is_in = epi == VITK_EPI_GELU_BWD || epi == VITK_EPI_MUL_AUX || epi == VITK_EPI_MUL_AUX8;
            const bool aux_is_out = epi == VITK_EPI_BIAS_GELU || epi == VITK_EPI_BIAS_GELU_DG || epi == VITK_EPI_BIAS_GELU_DG8;
            const size_t aux_sz = (epi == VITK_EPI_BIAS_GELU_DG8 || epi == VITK_EPI_MUL_AUX8) ? 1 : 2;
            const bool has_bias = epi == VITK_EPI_BIAS || aux_is_out || epi == VITK_EPI_RESID16 || epi == VITK_EPI_RESID;
            const void* resid = epi == VITK_EPI_RESID ? r32.p : (epi == VITK_EPI_RESID16 ? r16.p : nullptr);
            const size_t cbytes = (size_t)M * N * (f32out ? 4 : 2);
            auto api = [&](Buf& C, Buf& X, Buf& cs) {
                void* aux = aux_is_in ? aux_in.p : (aux_is_out ? X.p : nullptr);
                if (epi == VITK_EPI_MUL_AUX) VK(vitk_gemm_nt_bf16_mul_aux_colsum(A.p, K, Wp.p, 0, C.p, N, M, N, K, aux, (float*)cs.p, nullptr));
                else if (epi == VITK_EPI_MUL_AUX8) VK(vitk_gemm_nt_bf16_mul_aux8_colsum(A.p, K, Wp.p, 0, C.p, N, M, N, K, aux, (float*)cs.p, nullptr));
                else if (epi == VITK_EPI_GELU_BWD) VK(vitk_gemm_nt_bf16_gelu_bwd_colsum(A.p, K, Wp.p, 0, C.p, N, M, N, K, aux, (float*)cs.p, nullptr));
                else VK(vitk_gemm_nt_bf16(A.p, K, Wp.p, 0, C.p, N, M, N, K, epi, has_bias ? bias.p : nullptr, (const float*)resid, aux, nullptr));
            };
            auto direct = [&](int tiles_m, int abl, int dbg) {      // the four-wave launch alone
                void* aux = aux_is_in ? aux_in.p : (aux_is_out ? X1.p : nullptr);
                VK(probe_ntw_launch(tiles_m, grid, A.p, K, Wp.p, 0, C1.p, N, N, K, epi, has_bias ? bias.p : nullptr, resid, aux,
                                    aux_is_in ? (float*)cs1.p : nullptr, abl, dbg, nullptr));
            };
            // ---- parity: product path vs the 8-wave kernel alone, bit for bit ----
            CK(hipMemset(C0.p, 0xff, cbytes)); CK(hipMemset(C1.p, 0xee, cbytes));
            CK(hipMemset(X0.p, 0xff, M * N * 2)); CK(hipMemset(X1.p, 0xee, M * N * 2));
            CK(hipMemset(cs0.p, 0, cs0.n)); CK(hipMemset(cs1.p, 0, cs1.n));
            setenv("VITK_NT_W128", "0", 1);
            const int64_t rows_old = vitk_gemm_nt_colsum_rows(M, N, K, N);
            api(C0, X0, cs0);
            unsetenv("VITK_NT_W128");
            const int64_t rows_new = vitk_gemm_nt_colsum_rows(M, N, K, N);
            api(C1, X1, cs1);
            CK(hipDeviceSynchronize());
            std::vector<unsigned char> h0(cbytes), h1(cbytes);
            CK(hipMemcpy(h0.data(), C0.p, cbytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), C1.p, cbytes, hipMemcpyDeviceToHost));
            size_t bad = 0, first_bad = 0;
            for (size_t i = 0; i < cbytes; ++i) if (h0[i] != h1[i]) { if (!bad) first_bad = i; ++bad; }
            size_t badx = 0;
            if (aux_is_out) {
                std::vector<unsigned char> x0((size_t)M * N * aux_sz), x1((size_t)M * N * aux_sz);
                CK(hipMemcpy(x0.data(), X0.p, x0.size(), hipMemcpyDeviceToHost)); CK(hipMemcpy(x1.data(), X1.p, x1.size(), hipMemcpyDeviceToHost));
                for (size_t i = 0; i < x0.size(); ++i) badx += x0[i] != x1[i];
            }
            double cs_err = 0;
            if (aux_is_in) {      // column sums: the partial rows differ between the two plans, their sums agree to f32 round-off
                std::vector<float> p0((size_t)rows_old * N), p1((size_t)rows_new * N);
                CK(hipMemcpy(p0.data(), cs0.p, p0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(p1.data(), cs1.p, p1.size() * 4, hipMemcpyDeviceToHost));
                for (int64_t n = 0; n < N; ++n) {
                    double a = 0, b = 0;
                    for (int64_t r = 0; r < rows_old; ++r) a += p0[r * N + n];
                    for (int64_t r = 0; r < rows_new; ++r) b += p1[r * N + n];
                    cs_err = std::max(cs_err, fabs(a - b) / (fabs(a) + 1.0));
                }
            }
            printf("  %-13s parity: C %s (%zu bytes differ%s)", epi_name[epi], bad ? "MISMATCH" : "bit-identical", bad, bad ? "" : "");
            if (bad) printf(" first at byte %zu (row %zu col %zu)", first_bad, first_bad / (N * (f32out ? 4 : 2)), (first_bad % (N * (f32out ? 4 : 2))) / (f32out ? 4 : 2));
            if (aux_is_out) printf(", aux %s (%zu)", badx ? "MISMATCH" : "bit-identical", badx);
            if (aux_is_in) printf(", colsum rows %lld vs %lld, rel diff of the sums %.2e", (long long)rows_old, (long long)rows_new, cs_err);
            printf("\n");
            {   // round 5's feed must agree with round 6's, bit for bit (C only: same code beyond the feed)
                CK(hipMemset(C0.p, 0xdd, cbytes));
                setenv("VITK_NTW_A128", "0", 1);
                api(C0, X0, cs0);
                unsetenv("VITK_NTW_A128");
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(h0.data(), C0.p, cbytes, hipMemcpyDeviceToHost));
                size_t bad5 = 0;
                for (size_t i = 0; i < cbytes; ++i) bad5 += h0[i] != h1[i];
                printf("  %-13s 64-byte vs 128-byte activation feed: C %s (%zu bytes differ)\n", epi_name[epi], bad5 ? "MISMATCH" : "bit-identical", bad5);
            }
            // ---- A/B timing, interleaved rounds ----
            std::vector<float> t_old, t_new, t_dir, t_r5, t_dir5;
            for (int r = 0; r <= rounds; ++r) {
                setenv("VITK_NT_W128", "0", 1);
                const float a = time_ms([&] { api(C0, X0, cs0); }, 10);
                unsetenv("VITK_NT_W128");
                const float b = time_ms([&] { api(C1, X1, cs1); }, 10);
                const float c = time_ms([&] { direct(tm_all, 0, 0); }, 10);
                setenv("VITK_NTW_A128", "0", 1);        // round 5's feed: the activation operand as 64-byte row pieces
                const float d = time_ms([&] { api(C1, X1, cs1); }, 10);
                const float e = time_ms([&] { direct(tm_all, 0, 0); }, 10);
                unsetenv("VITK_NTW_A128");
                if (r) { t_old.push_back(a); t_new.push_back(b); t_dir.push_back(c); t_r5.push_back(d); t_dir5.push_back(e); }
            }
            const float mo = median(t_old), mn = median(t_new), md = median(t_dir), m5 = median(t_r5), md5 = median(t_dir5);
            printf("  %-13s r4 8-wave %7.1f us %7.1f TF/s | product path %7.1f us %7.1f TF/s (x%.3f) | four-wave kernel alone, all %d m-tiles %7.1f us %7.1f TF/s\n",
                   epi_name[epi], mo * 1e3, flop / mo / 1e9, mn * 1e3, flop / mn / 1e9, mo / mn, tm_all, md * 1e3, 2.0 * 256 * tm_all * N * K / md / 1e9);
            printf("  %-13s    with round 5's 64-byte activation pieces (VITK_NTW_A128=0): product path %7.1f us (128-byte rows x%.3f) | four-wave kernel alone %7.1f us (x%.3f)\n",
                   epi_name[epi], m5 * 1e3, m5 / mn, md5 * 1e3, md5 / md);
            if (epi == sh.epis[0] || epi == VITK_EPI_RESID16 || epi == VITK_EPI_BIAS_GELU_DG8 || epi == VITK_EPI_MUL_AUX8) { sum_old += mo; sum_new += mn; }
            // ---- round 6: two wave groups of one workgroup swapping roles (csrc/gemm_nt_x2.hip): parity against the four-wave kernel alone, timing ----
            const bool x2_epi = epi == VITK_EPI_NONE || epi == VITK_EPI_BIAS_GELU_DG || epi == VITK_EPI_BIAS_GELU_DG8 || epi == VITK_EPI_MUL_AUX || epi == VITK_EPI_MUL_AUX8;
            if (x2_epi && probe_ntx_serves(M, N, K)) {
                auto x2 = [&](int abl, int dbg) {
                    void* aux = aux_is_in ? aux_in.p : (aux_is_out ? X0.p : nullptr);
                    VK(probe_ntx_launch(tm_all, grid, A.p, K, Wp.p, 0, C0.p, N, N, K, epi, has_bias ? bias.p : nullptr, resid, aux,
                                        aux_is_in ? (float*)cs0.p : nullptr, abl, dbg, nullptr));
                };
                const size_t cb = (size_t)256 * tm_all * N * 2;
                CK(hipMemset(C0.p, 0xdd, cb)); CK(hipMemset(C1.p, 0xee, cb));
                CK(hipMemset(X0.p, 0xdd, M * N * 2)); CK(hipMemset(X1.p, 0xee, M * N * 2));
                CK(hipMemset(cs0.p, 0, cs0.n)); CK(hipMemset(cs1.p, 0, cs1.n));
                x2(0, 0);
                direct(tm_all, 0, 0);
                CK(hipDeviceSynchronize());
                std::vector<unsigned char> g0(cb), g1(cb);
                CK(hipMemcpy(g0.data(), C0.p, cb, hipMemcpyDeviceToHost)); CK(hipMemcpy(g1.data(), C1.p, cb, hipMemcpyDeviceToHost));
                size_t bx = 0, fbx = 0;
                for (size_t i = 0; i < cb; ++i) if (g0[i] != g1[i]) { if (!bx) fbx = i; ++bx; }
                size_t bax = 0, bcs = 0;
                if (aux_is_out) {
                    const size_t ab = (size_t)256 * tm_all * N * aux_sz;
                    std::vector<unsigned char> y0(ab), y1(ab);
                    CK(hipMemcpy(y0.data(), X0.p, ab, hipMemcpyDeviceToHost)); CK(hipMemcpy(y1.data(), X1.p, ab, hipMemcpyDeviceToHost));
                    for (size_t i = 0; i < ab; ++i) bax += y0[i] != y1[i];
                }
                if (aux_is_in) {
                    const size_t sb = (size_t)2 * tm_all * N * 4;
                    std::vector<unsigned char> y0(sb), y1(sb);
                    CK(hipMemcpy(y0.data(), cs0.p, sb, hipMemcpyDeviceToHost)); CK(hipMemcpy(y1.data(), cs1.p, sb, hipMemcpyDeviceToHost));
                    for (size_t i = 0; i < sb; ++i) bcs += y0[i] != y1[i];
                }
                printf("  %-13s TWO GROUPS (x2) vs the four-wave kernel alone: C %s (%zu bytes differ", epi_name[epi], bx ? "MISMATCH" : "bit-identical", bx);
                if (bx) printf(", first at row %zu col %zu", fbx / (N * 2), (fbx % (N * 2)) / 2);
                printf(")");
                if (aux_is_out) printf(", second output %s (%zu)", bax ? "MISMATCH" : "bit-identical", bax);
                if (aux_is_in) printf(", column-sum rows %s (%zu)", bcs ? "MISMATCH" : "bit-identical", bcs);
                printf("\n");
                std::vector<float> tx, tw, txm;
                for (int r = 0; r <= rounds; ++r) {
                    const float a2 = time_ms([&] { x2(0, 0); }, 10);
                    const float b2 = time_ms([&] { direct(tm_all, 0, 0); }, 10);
                    const float c2 = time_ms([&] { x2(0, 1); }, 10);
                    if (r) { tx.push_back(a2); tw.push_back(b2); txm.push_back(c2); }
                }
                printf("  %-13s TWO GROUPS (x2) %7.1f us %7.1f TF/s | four-wave kernel alone %7.1f us (x%.3f) | x2 main loops alone (no epilogue pieces) %7.1f us\n",
                       epi_name[epi], median(tx) * 1e3, 2.0 * 256 * tm_all * N * K / median(tx) / 1e9, median(tw) * 1e3, median(tw) / median(tx), median(txm) * 1e3);
                if (aux_is_out) {
                    std::vector<float> t1, t2, t3;
                    for (int r = 0; r < rounds; ++r) { t1.push_back(time_ms([&] { x2(0, 32); }, 10)); t2.push_back(time_ms([&] { x2(0, 64); }, 10)); t3.push_back(time_ms([&] { x2(0, 96); }, 10)); }
                    printf("  %-13s x2 without the GELU arithmetic %7.1f us | without the stores %7.1f us | without both %7.1f us\n", epi_name[epi], median(t1) * 1e3, median(t2) * 1e3, median(t3) * 1e3);
                }
                if (epi == VITK_EPI_NONE) {
                    std::vector<float> t1, t2;
                    for (int r = 0; r < rounds; ++r) { t1.push_back(time_ms([&] { x2(1, 1); }, 10)); t2.push_back(time_ms([&] { direct(tm_all, 1, 1); }, 10)); }
                    printf("  %-13s x2 main loops WITHOUT LDS-DMA (reads + MFMA + barriers) %7.1f us | four-wave kernel the same %7.1f us\n", epi_name[epi], median(t1) * 1e3, median(t2) * 1e3);
                }
            }
            if (quick) continue;
            // ---- ablations of the four-wave launch (all full tiles) ----
            if (epi == VITK_EPI_NONE) {
                struct Ab { const char* name; int abl, dbg; };
                const Ab abs[] = {{"all", 0, 0}, {"strict waits after the epilogue", 0, 2}, {"main loop alone (no epilogue)", 0, 1},
                                  {"LDS-DMA only", 6, 1}, {"MFMA only", 3, 1}, {"fragment reads only", 5, 1}, {"DMA + MFMA", 2, 1}, {"reads + MFMA", 1, 1},
                                  {"empty loop", 7, 1}, {"epilogue only (empty loop + stores)", 7, 0},
                                  {"32x32x16 MFMAs (timing only): main loop", 8, 1}, {"32x32x16 MFMAs: DMA + MFMA", 10, 1}, {"32x32x16 MFMAs: MFMA only", 11, 1}};
                for (const Ab& ab : abs) {
                    std::vector<float> t, t5;
                    for (int r = 0; r < rounds; ++r) {
                        t.push_back(time_ms([&] { direct(tm_all, ab.abl, ab.dbg); }, 10));
                        setenv("VITK_NTW_A128", "0", 1);
                        t5.push_back(time_ms([&] { direct(tm_all, ab.abl, ab.dbg); }, 10));
                        unsetenv("VITK_NTW_A128");
                    }
                    printf("      %-42s %7.1f us   (64-byte activation pieces: %7.1f us)\n", ab.name, median(t) * 1e3, median(t5) * 1e3);
                }
                // the same launches on ZERO operands: same instruction stream and addresses, no toggling in the matrix cores / data paths -- what
                // the chip's power management (DVFS) takes from the random-data numbers above
                Buf Az = dalloc(M * K * 2), Wz = dalloc(vitk_pack_w_nt_bytes(N, K));
                CK(hipMemset(Az.p, 0, Az.n)); CK(hipMemset(Wz.p, 0, Wz.n));
                auto direct_z = [&](int abl, int dbg) {
                    VK(probe_ntw_launch(tm_all, grid, Az.p, K, Wz.p, 0, C1.p, N, N, K, epi, nullptr, nullptr, nullptr, nullptr, abl, dbg, nullptr));
                };
                const Ab zs[] = {{"ZERO operands: all", 0, 0}, {"ZERO operands: main loop alone", 0, 1}, {"ZERO operands: MFMA only", 3, 1}, {"ZERO operands: DMA + MFMA", 2, 1},
                                 {"ZERO operands: 32x32x16 main loop", 8, 1}};
                for (const Ab& ab : zs) {
                    std::vector<float> t, t5;
                    for (int r = 0; r < rounds; ++r) {
                        t.push_back(time_ms([&] { direct_z(ab.abl, ab.dbg); }, 10));
                        setenv("VITK_NTW_A128", "0", 1);
                        t5.push_back(time_ms([&] { direct_z(ab.abl, ab.dbg); }, 10));
                        unsetenv("VITK_NTW_A128");
                    }
                    printf("      %-42s %7.1f us   (64-byte activation pieces: %7.1f us)\n", ab.name, median(t) * 1e3, median(t5) * 1e3);
                }
                CK(hipFree(Az.p)); CK(hipFree(Wz.p));
            } else if (epi == VITK_EPI_BIAS_GELU_DG || epi == VITK_EPI_MUL_AUX || epi == VITK_EPI_RESID16 || epi == VITK_EPI_BIAS_GELU_DG8 || epi == VITK_EPI_MUL_AUX8) {
                struct Ab { const char* name; int abl, dbg; };
                const Ab abs[] = {{"epilogue only (empty loop + epilogue)", 7, 0}, {"strict waits after the epilogue", 0, 2},
                                  {"epilogue only, no GELU arithmetic (DG epilogues)", 7, 32}, {"epilogue only, no stores (DG / MUL_AUX epilogues)", 7, 64},
                                  {"all, no GELU arithmetic (DG epilogues)", 0, 32}, {"all, no stores (DG / MUL_AUX epilogues)", 0, 64},
                                  {"main loop alone (no epilogue)", 0, 1}};
                for (const Ab& ab : abs) {
                    if ((ab.dbg & 96) && !aux_is_out) continue;      // the GELU / store ablations exist in the DG epilogues (exact-count waits elsewhere)
                    std::vector<float> t, t5;
                    for (int r = 0; r < rounds; ++r) {
                        t.push_back(time_ms([&] { direct(tm_all, ab.abl, ab.dbg); }, 10));
                        setenv("VITK_NTW_A128", "0", 1);
                        t5.push_back(time_ms([&] { direct(tm_all, ab.abl, ab.dbg); }, 10));
                        unsetenv("VITK_NTW_A128");
                    }
                    printf("      %-42s %7.1f us   (round 5's feed and epilogue: %7.1f us)\n", ab.name, median(t) * 1e3, median(t5) * 1e3);
                }
            }
        }
        for (Buf* b : {&A, &W, &bias, &aux_in, &r16, &r32, &Wp, &C0, &C1, &X0, &X1, &cs0, &cs1}) CK(hipFree(b->p));
    }
    printf("\nsum over the headline epilogues of the shapes above: 8-wave %.3f ms, product path %.3f ms (x%.3f)\n", sum_old, sum_new, sum_old / sum_new);
    return 0;
}
