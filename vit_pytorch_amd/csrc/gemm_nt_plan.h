// gemm_nt_plan.h -- tile schedule of the persistent NT GEMM (gemm_nt_persist.hip), shared with gemm_bf16.hip (dispatch,
// vitk_gemm_nt_colsum_rows).
//
// The output is cut into a MAIN region of 256-row tiles (rows [0, 256 * tm_main)) and a TAIL region of 128-row tiles (the
// remaining rows), both 256 columns wide.  One workgroup per CU stays resident and walks a static list of tiles; the tail
// height exists because the grid is quantised in rounds of 256 tiles: ViT-B/16 at batch 256 has M = 50,432 rows, i.e. 197
// m-tiles x 3 n-tiles = 591 tiles for N = 768 = 2.31 rounds that cost 3.  With the last 79 tile-equivalents re-cut into 158
// half tiles the last round costs half a round.  The plan (how many rows go to the tail) is chosen on the host by
// evaluating the makespan of the static assignment under a two-parameter cost model.
#pragma once
#include <stdint.h>

struct NtpPlan {
    bool ok;            // shape served by the persistent kernel
    int tiles_n;        // 256-column tiles
    int group_n;        // n-tiles per group of the grouped tile order
    int tm_main;        // 256-row m-tiles of the main region
    int tail_tm;        // 128-row m-tiles of the tail region
    int n_main, n_tail; // tile counts (tm_main * tiles_n, tail_tm * tiles_n)
    int nt;             // K-steps of 32 (16-bit operands) per tile
    int grid;           // resident workgroups (multiple of 8)
};

NtpPlan ntp_plan(int64_t M, int64_t N, int64_t K, int64_t ldc, const void* aux);
static inline int64_t ntp_colsum_rows(const NtpPlan& pl) { return 2 * ((int64_t)pl.tm_main + pl.tail_tm); }

int gemm_ntp_launch(const NtpPlan& pl, const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M,
                    int64_t N, int64_t K, int epilogue, const void* bias, const float* resid, void* aux, float* csum, unsigned drop_t, unsigned drop_seed,
                    float inv_keep, void* stream, unsigned drop_m0 = 0);

// the four-wave kernel (gemm_nt_w128.hip): full 256-row tiles only, rows [0, 256 tiles_m); `resid` is f32 (EPI_RESID) or 16-bit (EPI_RESID16)
bool gemm_ntw_serves(int64_t M, int64_t N, int64_t K);
int gemm_ntw_split(int64_t M, int64_t N, int64_t K, int grid);
int gemm_ntw_launch(int tiles_m, int grid, const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                    int64_t N, int64_t K, int epilogue, const void* bias, const void* resid, void* aux, float* csum, int abl, int dbg, void* stream);
