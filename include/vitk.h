/* vitk.h -- C-ABI of libvitk.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * ViT / SimpleViT encoder forward + backward hot path.
 *
 * The reference (lucidrains/vit-pytorch) has NO native layer and NO FFI: its hot path is
 * Python calling torch.nn / einops.  The "FFI the reference would bind" is
 * therefore the set of ATen ops its hot path dispatches; each entry point below names the
 * reference call site(s) (file:line under /root/reference/vit_pytorch/) it replaces.
 *
 * Conventions
 *   - plain C: raw device pointers, int64 sizes, a dtype tag, and the hipStream_t (passed as
 *     void*) to enqueue on.  No torch types.  Nothing here allocates, frees or synchronises;
 *     kernels are enqueued on `stream` and the call returns.  Never uses the null stream
 *     unless the caller passes it.
 *   - return value: 0 = OK; < 0 = argument/shape/alignment error detected on the host before
 *     any launch (VITK_E_*); > 0 = hipError_t from the launch.  vitk_last_error() returns a
 *     thread-local human readable string for the last non-zero return on this thread.
 *   - thread safety: re-entrant; the error string is thread-local.  Process-wide mutable state is limited to (1) the CU reserve
 *     (vitk_set_cu_reserve: an atomic int; it changes how many workgroups the large GEMMs are LAUNCHED on and the split count
 *     vitk_gemm_tn_splits() proposes -- never the layout of an output or of the column-sum partial rows, so a concurrent change
 *     costs speed, not correctness), (2) the RCCL communicator behind vitk_comm_* (guarded by a mutex), and (3) the ticket words of
 *     the dynamic tile order (one slot per launch, atomically claimed).  Callable from the autograd worker thread.
 *   - dtype tags: VITK_F32 = 0, VITK_BF16 = 1.  "T" below means the model dtype.  The same ABI ships twice:
 *     libvitk.so, where the 16-bit type is bfloat16, and libvitk_f16.so (same sources, -DVITK_HALF_IS_F16),
 *     where every "bf16" of this header is IEEE binary16 (model.half()): tag 1 then denotes half, the `_bf16` entry
 *     points run v_mfma_f32_16x16x32_f16, accumulation stays f32.  vitk_half_type() tells the two apart.
 *   - all matrices are row-major with explicit leading dimensions (in elements).
 */
#ifndef VITK_H
#define VITK_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VITK_VERSION 138

#define VITK_F32 0
#define VITK_BF16 1          /* the library's 16-bit float type: bfloat16 (libvitk.so) or IEEE half (libvitk_f16.so) */
#define VITK_F16 2           /* only as the return value of vitk_half_type() */

#define VITK_E_ARG (-1)      /* null pointer / bad enum */
#define VITK_E_SHAPE (-2)    /* unsupported extent */
#define VITK_E_ALIGN (-3)    /* pointer or leading dimension not aligned as required */
#define VITK_E_DTYPE (-4)    /* unsupported dtype combination */
#define VITK_E_UNAVAILABLE (-5)  /* an optional component is not present (vitk_comm_*: librccl could not be opened) */
#define VITK_E_COMM (-6)     /* RCCL reported an error (vitk_last_error() carries its text) */

int vitk_version(void);
int vitk_half_type(void);    /* VITK_BF16 or VITK_F16: what dtype tag 1 means in this library */
const char* vitk_last_error(void);

/* A row map sends logical row r to physical row (r / group) * gstride + (r % group) + offset.
 * Identity: group = 0 (treated as "no map").  Used to read the cls rows of a (B,N,D) tensor
 * (group 1, gstride N) or to write patch rows behind a cls slot (group Np, gstride N, offset 1). */
typedef struct vitk_rowmap { int64_t group, gstride, offset; } vitk_rowmap;

/* ------------------------------------------------------------------------------------------
 * LayerNorm  (replaces nn.LayerNorm at vit.py:19,39,69,101,103 / simple_vit.py:29,42,67,92,94)
 *   y[omap(r), :] = (x[imap(r), :] - mean_r) * rstd_r * w + b  (+ add[(r % add_group) + add_off, :])
 *   mean/rstd (f32, indexed by logical r) are saved for backward.  Biased variance, eps in sqrt.
 *   `add` (nullable, dtype wdt, ld = D) fuses the positional-embedding add of vit.py:127 /
 *   simple_vit.py:114.  D % 4 == 0, D <= 4096.
 * ------------------------------------------------------------------------------------------ */
int vitk_layernorm_fwd(const void* x, int xdt, const void* w, const void* b, int wdt,
                       void* y, int ydt, float* mean, float* rstd,
                       int64_t rows, int64_t D, float eps,
                       vitk_rowmap imap, vitk_rowmap omap,
                       const void* add, int64_t add_group, int64_t add_off, void* stream);

/* LayerNorm backward.
 *   g   = dy[dymap(r)] * w ; xhat = (x[xmap(r)] - mean_r) * rstd_r
 *   dx  = rstd_r * (g - mean(g) - xhat * mean(g * xhat))            (+ gin[dxmap(r)] if gin)
 *   written to dx_f32[dxmap(r)] and/or dx_T[dxmap(r)] (either may be null; both null = only dw/db)
 *   dw/db partial column sums go to `partials` (f32, 2 * nblk * D, nblk = vitk_layernorm_bwd_blocks(rows, D));
 *   finish with vitk_colsum_partials.  If colsum_dx != 0 a third slab (column sums of the value
 *   written to dx, i.e. the bias gradient of the Linear that produced the residual branch) is
 *   also accumulated: partials has 3 * nblk * D floats.
 */
int64_t vitk_layernorm_bwd_blocks(int64_t rows, int64_t D);
int vitk_layernorm_bwd(const void* dy, int dydt, const void* x, int xdt, const void* w, int wdt,
                       const float* mean, const float* rstd,
                       const float* gin, float* dx_f32, void* dx_t, int dxtdt,
                       float* partials, int colsum_dx,
                       int64_t rows, int64_t D,
                       vitk_rowmap dymap, vitk_rowmap xmap, vitk_rowmap dxmap, void* stream);
/* The same when the Linear that produced the residual branch ended in nn.Dropout(p) (vit.py:24,48): dx_t and the column
 * sums (= the gradient at that Linear's output and its bias gradient) carry the keep decisions / 1/(1-p) of (p, seed) at
 * (row, column); dx_f32, the residual stream gradient, is not masked.                                               */
int vitk_layernorm_bwd_drop(const void* dy, int dydt, const void* x, int xdt, const void* w, int wdt,
                            const float* mean, const float* rstd, const float* gin, float* dx_f32, void* dx_t, int dxtdt,
                            float* partials, int colsum_dx, int64_t rows, int64_t D, vitk_rowmap dymap, vitk_rowmap xmap,
                            vitk_rowmap dxmap, float drop_p, uint32_t drop_seed, void* stream);
/* The same with the stream gradient kept in the 16-bit type (what torch autograd does when the reference runs in bfloat16 --
 * vit.py:80-81's `+ x` in the backward): gin (may be null) and dx_t are both of dtype dxtdt, dx_t[dxmap(r)] = dx + gin[dxmap(r)] IS
 * the updated stream and the operand of the following weight-gradient / input-gradient GEMMs; there is no f32 output.  Moves
 * 386 MB instead of 619 MB per launch at 50,432 x 768.  16-bit parameters, D % 4 == 0, no dropout.                     */
int vitk_layernorm_bwd_s16(const void* dy, int dydt, const void* x, int xdt, const void* w, int wdt, const float* mean,
                           const float* rstd, const void* gin, void* dx_t, int dxtdt, float* partials, int colsum_dx,
                           int64_t rows, int64_t D, vitk_rowmap dymap, vitk_rowmap xmap, vitk_rowmap dxmap, void* stream);

/* One-launch finish of vitk_layernorm_bwd: dw, db (dtype odt, either may be null) and, if non-null, the f32
 * column sums dcol of dx, from the `partials` buffer of that call (nblk = vitk_layernorm_bwd_blocks(rows, D)). */
int vitk_layernorm_bwd_finalize(const float* partials, int64_t nblk, int64_t D, void* dw, void* db, int odt,
                                float* dcol, void* stream);
/* The same with the column sums of dx written in the parameter dtype when dcol_dt == odt (float32 when dcol_dt == VITK_F32): they
 * ARE the bias gradient of the Linear in front of the LayerNorm (autograd of vit.py:23,47), so the engine lets this kernel write
 * them straight into that gradient's buffer instead of casting a float32 temporary.                                             */
int vitk_layernorm_bwd_finalize_ex(const float* partials, int64_t nblk, int64_t D, void* dw, void* db, int odt,
                                   void* dcol, int dcol_dt, void* stream);

/* out[c] (dtype odt) = (accumulate ? out[c] : 0) + sum_{p < nparts} partials[p * ld + c], c < cols */
int vitk_colsum_partials(const float* partials, int64_t nparts, int64_t ld, int64_t cols,
                         void* out, int odt, int accumulate, void* stream);

/* Column sums of a (rows x cols) matrix (bias gradients: db = colsum(dY); pos/cls gradients:
 * sum over the batch of a (B, N*D) view).  ws: f32 workspace of vitk_colsum_ws_floats() floats. */
int64_t vitk_colsum_ws_floats(int64_t rows, int64_t cols);
int vitk_colsum(const void* x, int xdt, int64_t rows, int64_t cols, int64_t ld,
                void* out, int odt, int accumulate, float* ws, void* stream);

/* ------------------------------------------------------------------------------------------
 * bf16 MFMA GEMMs (replace nn.Linear at vit.py:20,23,44,47,102 and their autograd)
 * ------------------------------------------------------------------------------------------ */
#define VITK_EPI_NONE 0        /* C = acc                                  (C: bf16)            */
#define VITK_EPI_BIAS 1        /* C = acc + bias[n]                                             */
#define VITK_EPI_BIAS_GELU 2   /* aux = acc + bias (pre-activation, bf16); C = gelu_erf(aux)    */
#define VITK_EPI_RESID 3       /* Cf32 = resid_f32 + acc (+ bias[n] if bias)  (C: f32)          */
#define VITK_EPI_RESID16 5     /* C16 = T(resid16 + acc (+ bias[n])): the `+ x` of vit.py:80-81 with the forward residual stream in the
                                  16-bit type (opt-in, VITK_FWD_STREAM=16); `resid` then points at 16-bit data; persistent kernel only */
#define VITK_EPI_GELU_BWD 4    /* C = acc * gelu'(aux[m][n])   (aux = saved pre-activation)     */
#define VITK_EPI_BIAS_GELU_DG 6 /* BIAS_GELU whose aux output is gelu'(pre) (of the rounded pre-activation) instead of pre: what the backward needs of
                                  pre is only that factor, and with it stored the backward GEMM's epilogue is a multiplication (VITK_EPI_MUL_AUX)
                                  instead of a second polynomial + exponential per element; persistent kernel only, no fused dropout */
#define VITK_EPI_MUL_AUX 7     /* C = acc * aux[m][n]  (aux = the gelu' factor BIAS_GELU_DG stored); column sums as with GELU_BWD
                                  (vitk_gemm_nt_bf16_mul_aux_colsum); persistent kernel only */
#define VITK_EPI_BIAS_GELU_DG8 8 /* BIAS_GELU_DG with the factor stored in 8 bits: aux is M x ldc BYTES, code = rne(200 gelu'(pre)) + 27 (gelu' lies in
                                  [-0.129, 1.129]: codes 1 .. 253), i.e. fixed point with |error| <= 0.0025 -- a bf16 of the same value is off by up to
                                  0.0039 in [1, 1.13) -- and 0, 0.5, 1 exact; halves the bytes of FF1's second output and of dFF1's second input, both
                                  of which run at the memory system's rate; persistent kernel only, ldc % 8 == 0 */
#define VITK_EPI_MUL_AUX8 9    /* C = acc * 0.005 (aux8[m][n] - 27): VITK_EPI_MUL_AUX on the codes BIAS_GELU_DG8 stored (vitk_gemm_nt_bf16_mul_aux8_colsum) */

/* C[M,N] = A[M,K] . W[N,K]^T with a fused epilogue.  A, W bf16, K-contiguous ("NT").
 * Requirements: K % 32 == 0; lda, ldw % 8 == 0; N % 4 == 0; pointers 16-byte aligned.
 * bias: bf16 [N] or null.  resid: f32 [M, ldc].  aux: bf16 [M, ldc].                          */
int vitk_gemm_nt_bf16(const void* A, int64_t lda, const void* W, int64_t ldw,
                      void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                      int epilogue, const void* bias, const float* resid, void* aux, void* stream);

/* The VITK_EPI_GELU_BWD product with the column sums of C as a by-product (C = dY.W2 * gelu'(pre) is the gradient at
 * the first FeedForward Linear's output, so colsum(C) is that Linear's bias gradient -- autograd of vit.py:20).
 * Writes R = vitk_gemm_nt_colsum_rows(M, N, K, ldc) rows of N float partial sums (row-major, ld = N) taken over the
 * bf16-rounded C; fold them with vitk_colsum_partials(partials, R, N, N, ...).  R == 0: the shape is not served by
 * the 256-row kernel and this entry point refuses it (use vitk_gemm_nt_bf16 + vitk_colsum).                       */
int64_t vitk_gemm_nt_colsum_rows(int64_t M, int64_t N, int64_t K, int64_t ldc);
/* The same with VITK_EPI_MUL_AUX: C = (A . W^T) * aux, aux = the gelu' factor a VITK_EPI_BIAS_GELU_DG forward stored. */
int vitk_gemm_nt_bf16_mul_aux_colsum(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                                     int64_t M, int64_t N, int64_t K, const void* aux, float* colsum_partials, void* stream);
/* The same on the 8-bit codes of a VITK_EPI_BIAS_GELU_DG8 forward (aux8: M x ldc bytes). */
int vitk_gemm_nt_bf16_mul_aux8_colsum(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                                      int64_t M, int64_t N, int64_t K, const void* aux8, float* colsum_partials, void* stream);
/* Tile schedule of the persistent NT kernel for (M, N, K) (tests, tuning): out[0] = 1 when the shape is served by it,
 * out[1] = 256-row m-tiles, out[2] = 128-row m-tiles of the tail region, out[3] = resident workgroups, out[4] = n-tiles. */
int vitk_gemm_nt_plan(int64_t M, int64_t N, int64_t K, int64_t ldc, int32_t* out5);
/* K-blocked copy of a weight for the persistent NT kernel (no reference counterpart: nn.Linear's weight, vit.py:20,23,44,47,
 * stays as it is; this is a derived cache like the transposed copy).  W is N x K row-major 16-bit.  `out` receives the operand
 * of y = x . W^T (rows N, reduction K; K % 32 == 0; vitk_pack_w_nt_bytes(N, K) bytes), `out_t` the operand of dX = dY . W
 * (rows K, reduction N; N % 32 == 0; vitk_pack_w_nt_bytes(K, N) bytes); either may be null.  Pass the packed buffer to
 * vitk_gemm_nt_bf16 / _gelu_bwd_colsum / _drop as W with ldw == 0; only shapes vitk_gemm_nt_plan() reports as served accept
 * it (anything else fails with VITK_E_SHAPE).  Layout: for every (256-row tile, 32-element K-step) the 16 KiB image of the
 * kernel's LDS stage, so its LDS-DMA reads whole 128-byte lines (see gemm_nt_persist.hip).                                    */
int64_t vitk_pack_w_nt_bytes(int64_t rows, int64_t reduction);
int vitk_pack_w_nt(const void* W, int64_t ldw, int64_t N, int64_t K, void* out, void* out_t, void* stream);
/* The same for a table of `count` weights in ceil(jobs / 96) launches (a job = one non-null out / out_t): what a training step does
 * once per optimizer step for every Linear of the stack.  Row t: W[t] (N[t] x K[t], leading dimension ldw[t]), out[t], out_t[t]. */
int vitk_pack_w_nt_many(const void* const* W, const int64_t* ldw, const int64_t* N, const int64_t* K, void* const* out,
                        void* const* out_t, int64_t count, void* stream);
int vitk_gemm_nt_bf16_gelu_bwd_colsum(const void* A, int64_t lda, const void* W, int64_t ldw,
                                      void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                                      void* aux, float* colsum_partials, void* stream);
/* General form with fused nn.Dropout(p) (training mode; 256-row kernel only): RESID -- the Linear output is dropped before
 * the residual add (vit.py:24,48 with :80-81); BIAS_GELU -- C = dropout(gelu(aux)), aux stays undropped (vit.py:21-22);
 * GELU_BWD -- the matching backward factor; colsum_partials as in vitk_gemm_nt_bf16_gelu_bwd_colsum or null.  Keep decision
 * of element (m, n): hash(hash(m ^ seed) + n) >= p * 2^32 (vitk_dropout_keep).                                       */
int vitk_gemm_nt_bf16_drop(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M, int64_t N,
                           int64_t K, int epilogue, const void* bias, const float* resid, void* aux, float* colsum_partials,
                           float drop_p, uint32_t drop_seed, void* stream);

/* fp8 operands (BASELINE config 5): C = alpha * A8[M,K] . W8[N,K]^T with the epilogues of
 * vitk_gemm_nt_bf16; A8, W8 are OCP e4m3 bytes (K contiguous, K %% 64 == 0, lda / ldw %% 16 == 0), C / bias / aux the
 * library's 16-bit type, accumulation f32 on v_mfma_f32_16x16x32_fp8_fp8.  alpha = 1 / (scale_a * scale_w) undoes the
 * per-tensor scales.  Served by the 256-row kernel only (M >= 1024, N >= 256).
 * vitk_fp8_amax_scale: scale2[0] = 448 / max|x|, scale2[1] = max|x| / 448, computed on the device (no host sync).
 * vitk_quantize_fp8: out[i] = e4m3(clamp(x[i] * scale, +-448)), round to nearest even; scale from scale_dev[0] if given. */
int vitk_gemm_nt_fp8(const void* A8, int64_t lda, const void* W8, int64_t ldw, void* C, int64_t ldc, int64_t M, int64_t N,
                     int64_t K, int epilogue, const void* bias, const float* resid, void* aux, float alpha, void* stream);
int vitk_fp8_amax_scale(const void* x, int dt, int64_t n, float* scale2, void* stream);
int vitk_quantize_fp8(const void* x, int dt, void* out, int64_t n, const float* scale_dev, float scale_host, void* stream);
/* Delayed-scaling variants: producers emit the e4m3 copy themselves.
 * vitk_layernorm_fwd_fp8: vitk_layernorm_fwd that also writes y8 = e4m3(y * scale8[0]) (row-major like y; null = no copy)
 *   and records max|y| of this call into amax64 (64 words, float bit patterns, atomicMax; null = do not record).
 * vitk_gemm_nt_fp8_ex: A is e4m3 (a_is_fp8 != 0; W then too) or 16-bit; alpha_a / alpha_w are device pointers to inverse
 *   scales multiplied into alpha (null = 1); with the BIAS_GELU epilogue c8 / c8_scale / c8_amax64 give the activation an
 *   e4m3 copy and an amax record the same way.
 * vitk_fp8_update_scales: for every slot s < nslots: m = max over amax64[64 s .. 64 s + 63]; if m > 0:
 *   scales2[2 s] = 448 / m, scales2[2 s + 1] = m / 448; the words are reset to 0.                                     */
int vitk_layernorm_fwd_fp8(const void* x, int xdt, const void* w, const void* b, int wdt, void* y, int ydt,
                           float* mean, float* rstd, int64_t rows, int64_t D, float eps, vitk_rowmap imap, vitk_rowmap omap,
                           const void* add, int64_t add_group, int64_t add_off, void* y8, const float* scale8,
                           uint32_t* amax64, void* stream);
int vitk_gemm_nt_fp8_ex(const void* A, int64_t lda, int a_is_fp8, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M,
                        int64_t N, int64_t K, int epilogue, const void* bias, const float* resid, void* aux, float alpha,
                        const float* alpha_a, const float* alpha_w, void* c8, const float* c8_scale, uint32_t* c8_amax64,
                        void* stream);
int vitk_fp8_update_scales(uint32_t* amax64, float* scales2, int64_t nslots, void* stream);
/* fp8 for the BACKWARD and the out-projection (round 3; vit.py:20,23,44,47 -- the autograd of nn.Linear -- and vit.py:46-49):
 * vitk_gemm_nt_fp8_v2: vitk_gemm_nt_fp8_ex with (a) a_kind = 0 (A in the 16-bit type), 1 (e4m3) or 2 (OCP e5m2: gradients;
 *   W stays e4m3, v_mfma_f32_16x16x32_fp8_bf8; NONE and GELU_BWD epilogues -- dX = dY . W of the four Linear layers),
 *   (b) colsum_partials for the GELU_BWD epilogue (vitk_gemm_nt_fp8_colsum_rows(M, N, K, ldc) rows of N floats; fold with
 *   vitk_colsum_partials); with GELU_BWD c8 / c8_scale / c8_amax64 give the output an e5m2 copy and an amax record (the operand
 *   of the next dX / dW GEMMs); with BIAS_GELU, fp8 operands and a c8 output, C may be null (only the e4m3 copy is wanted),
 *   (c) flags bit 0: the K = 128 instruction (v_mfma_f32_16x16x128_f8f6f4 with unit block scales, the
 *   only fp8 form above the bf16 matrix rate on gfx950; K %% 128 == 0, 1-byte A).
 * vitk_quantize_fp8_delayed: ONE pass over a 16-bit / f32 tensor that (out8 != null) writes out8 = fp8(clamp(x * scale2[0]))
 *   -- fmt 0 = e4m3 (+-448), 1 = e5m2 (+-57344), round to nearest even -- under the scale decided BEFORE this step and
 *   (amax64 != null) records max|x| of THIS step into 64 words (float bit patterns, atomicMax), as the producers of
 *   vitk_layernorm_fwd_fp8 do.  n %% 4 == 0.
 * vitk_fp8_update_scales_fmt: vitk_fp8_update_scales with a per-slot format maximum fmax[s] (null = 448 everywhere):
 *   scales2[2 s] = fmax[s] / m, scales2[2 s + 1] = m / fmax[s].                                                        */
int vitk_gemm_nt_fp8_v2(const void* A, int64_t lda, int a_kind, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M,
                        int64_t N, int64_t K, int epilogue, const void* bias, const float* resid, void* aux,
                        float* colsum_partials, float alpha, const float* alpha_a, const float* alpha_w, void* c8,
                        const float* c8_scale, uint32_t* c8_amax64, int flags, void* stream);
int64_t vitk_gemm_nt_fp8_colsum_rows(int64_t M, int64_t N, int64_t K, int64_t ldc);
int vitk_quantize_fp8_delayed(const void* x, int dt, void* out8, int64_t n, const float* scale2, uint32_t* amax64, int fmt,
                              void* stream);
int vitk_fp8_update_scales_fmt(uint32_t* amax64, float* scales2, int64_t nslots, const float* fmax, void* stream);
/* Weight gradient on fp8 operands: dW[N,K] = alpha_y * alpha_x * sum_m dY8[m,N]^T X8[m,K] (autograd of nn.Linear, vit.py:20,23,44,47).
 * dY8: OCP e5m2 bytes (the copy vitk_quantize_fp8_delayed made for the dX GEMM), X8: OCP e4m3 bytes, both row-major with the
 * token index as the slow one (ldy, ldx %% 16 == 0); fragments by ds_read_b64_tr_b8, v_mfma_f32_16x16x32_bf8_fp8 or (flags bit 0)
 * v_mfma_f32_16x16x128_f8f6f4.  Split over M into `splits` f32 slabs (ws: splits * N * K floats; vitk_gemm_tn_fp8_splits, 0 =
 * shape not served: M >= 1024, N, K >= 256 and multiples of 16), folded into dW (dtype odt, ld = ldo; accumulate: dW += ...)
 * with the two device-resident inverse scales (null = 1).                                                              */
int64_t vitk_gemm_tn_fp8_splits(int64_t M, int64_t N, int64_t K, int flags);
int vitk_gemm_tn_fp8(const void* dY8, int64_t ldy, const void* X8, int64_t ldx, void* dW, int odt, int64_t ldo, int accumulate,
                     int64_t M, int64_t N, int64_t K, float* ws, int64_t splits, const float* alpha_y, const float* alpha_x,
                     int flags, void* stream);

/* dW[N,K] = sum_m dY[m,N]^T X[m,K]  ("TN": both operands are read with the reduction index as
 * the strided one).  Split over M into `splits` slabs of f32 partials (ws: splits*N*K floats),
 * then reduced into dW (dtype odt, ld = ldo; accumulate: dW += ...).  N % 8 == 0, K % 8 == 0.  */
int64_t vitk_gemm_tn_splits(int64_t M, int64_t N, int64_t K);
/* CUs the large GEMMs leave to other kernels (train_vit_decorr.py:74-78,109: the gradient all-reduce that overlaps the
 * backward): vitk_gemm_tn_splits plans for 256 - cus workgroups, the persistent NT kernels are launched on 256 - cus workgroups
 * (four-wave kernel) / draw their tiles by dynamic tickets (8-wave kernel).  Process-wide; 0 (default) = the whole chip.    */
int vitk_set_cu_reserve(int cus);
int vitk_get_cu_reserve(void);
/* Test hook: hold `ncus` CUs for `ms` milliseconds on `stream` (a stand-in for a collective's resident kernel).            */
int vitk_test_occupy_cus(int ncus, float ms, void* stream);
int vitk_gemm_tn_bf16(const void* dY, int64_t ldy, const void* X, int64_t ldx,
                      void* dW, int odt, int64_t ldo, int accumulate,
                      int64_t M, int64_t N, int64_t K, float* ws, int64_t splits, void* stream);
/* TWO weight gradients over the same M token rows in one launch (e.g. to_qkv's and to_out's, vit.py:44,47): their tiles share the split
 * count, so the pair fills the chip with fewer, larger M-splits -- fewer f32 slabs, one launch and one fold less.  Same operand rules as
 * vitk_gemm_tn_bf16 for each problem; odt = dtype of both outputs; ws: splits * (N0*K0 + N1*K1) floats;
 * splits = vitk_gemm_tn_pair_splits(...) -- 0 means "not served as a pair, call vitk_gemm_tn_bf16 twice".                         */
int64_t vitk_gemm_tn_pair_splits(int64_t M, int64_t N0, int64_t K0, int64_t N1, int64_t K1);
int vitk_gemm_tn_bf16_pair(const void* dY0, int64_t ldy0, const void* X0, int64_t ldx0, void* dW0, int64_t ldo0, int accumulate0,
                           int64_t N0, int64_t K0, const void* dY1, int64_t ldy1, const void* X1, int64_t ldx1, void* dW1,
                           int64_t ldo1, int accumulate1, int64_t N1, int64_t K1, int odt, int64_t M, float* ws, int64_t splits,
                           void* stream);

/* Generic strided batched GEMM for everything the fast kernels do not cover (f32 validation
 * mode, odd extents, the materialising attention path needed by forward hooks on `attend`):
 *   C[b1,b2][m][n] = alpha * sum_k A[b1,b2][m][k] * B[b1,b2][k][n] + beta * C (+ bias[n])
 * every operand addressed as base + b1*s_b1 + b2*s_b2 + row*s_row + col*s_col (element strides),
 * dtypes f32 or bf16 per operand, f32 accumulation.                                            */
typedef struct vitk_mat { const void* p; int dt; int64_t s_b1, s_b2, s_row, s_col; } vitk_mat;
int vitk_gemm_generic(vitk_mat A, vitk_mat B, vitk_mat C, const void* bias, int bias_dt,
                      int64_t nb1, int64_t nb2, int64_t M, int64_t N, int64_t K,
                      float alpha, float beta, void* stream);

/* ------------------------------------------------------------------------------------------
 * Attention core  (replaces vit.py:57-62: matmul(q,k^T)*scale -> softmax -> matmul(attn,v))
 * Fused, never materialises the N x N matrix.  bf16, dim_head == 64.
 * q/k/v/o addressed as base + b*s_b + h*s_h + n*s_n + d (element strides, d contiguous), so the
 * merged (B, N, 3*h*d) output of the to_qkv GEMM is consumed in place (no head-split copies:
 * replaces the three rearranges of vit.py:55 and the merge of vit.py:63).
 * lse: f32 (B, H, N) row log-sum-exp of the scaled scores, saved for backward.
 * ------------------------------------------------------------------------------------------ */
typedef struct vitk_bhnd { void* p; int64_t s_b, s_h, s_n; } vitk_bhnd;
int vitk_attn_fwd_bf16(vitk_bhnd q, vitk_bhnd k, vitk_bhnd v, vitk_bhnd o, float* lse,
                       int64_t B, int64_t H, int64_t N, int64_t d, float scale, void* stream);
/* delta: f32 (B,H,N) SCRATCH of the call, contents unspecified afterwards: the two-kernel path keeps rowsum(dO*O) in it between its
 * kernels; the single-kernel path (16-bit, 192 < N <= 208, no dropout, no CU reserve) keeps that in LDS and uses the buffer only as a
 * target for the store slots of rows that do not exist. */
int vitk_attn_bwd_bf16(vitk_bhnd q, vitk_bhnd k, vitk_bhnd v, vitk_bhnd o, vitk_bhnd dout,
                       const float* lse, float* delta,
                       vitk_bhnd dq, vitk_bhnd dk, vitk_bhnd dv,
                       int64_t B, int64_t H, int64_t N, int64_t d, float scale, void* stream);
/* The same with nn.Dropout(p) on the attention matrix (vit.py:42,60; training mode): kept probabilities are scaled by
 * 1 / (1 - p) for P.V while the softmax denominators stay those of the full row.  The keep decision of entry
 * (b, h, query, key) is hash(hash((b*H + h)*N + query ^ seed) + key) >= p * 2^32 -- recomputed by the backward kernels from
 * the same (p, seed), no mask tensor exists; vitk_dropout_keep() materialises the decisions for tests.              */
int vitk_attn_fwd_bf16_drop(vitk_bhnd q, vitk_bhnd k, vitk_bhnd v, vitk_bhnd o, float* lse, int64_t B, int64_t H,
                            int64_t N, int64_t d, float scale, float drop_p, uint32_t drop_seed, void* stream);
int vitk_attn_bwd_bf16_drop(vitk_bhnd q, vitk_bhnd k, vitk_bhnd v, vitk_bhnd o, vitk_bhnd dout, const float* lse,
                            float* delta, vitk_bhnd dq, vitk_bhnd dk, vitk_bhnd dv, int64_t B, int64_t H, int64_t N,
                            int64_t d, float scale, float drop_p, uint32_t drop_seed, void* stream);
/* f32-ACCURATE flavour of the same kernels, for the f32 validation mode (round 3; vit.py:55-63 again): every 16-bit operand
 * arrives as TWO tensors hi + lo (vitk_split2: hi = round16(x), lo = round16(x - hi)), every product keeps hi.hi + hi.lo + lo.hi,
 * the probabilities / dS are split the same way inside the kernel, outputs (o, dq, dk, dv: element strides of FLOAT tensors) are
 * f32.  Same staging, pipelining, masking and softmax code as vitk_attn_fwd/bwd_bf16 at 32 < N <= 224, dim_head 64.          */
int vitk_split2(const float* x, void* hi, void* lo, int64_t n, void* stream);
int vitk_attn_fwd_x2(vitk_bhnd q_hi, vitk_bhnd q_lo, vitk_bhnd k_hi, vitk_bhnd k_lo, vitk_bhnd v_hi, vitk_bhnd v_lo, vitk_bhnd o_f32,
                     float* lse, int64_t B, int64_t H, int64_t N, int64_t d, float scale, void* stream);
int vitk_attn_bwd_x2(vitk_bhnd q_hi, vitk_bhnd q_lo, vitk_bhnd k_hi, vitk_bhnd k_lo, vitk_bhnd v_hi, vitk_bhnd v_lo, vitk_bhnd o_f32,
                     vitk_bhnd do_hi, vitk_bhnd do_lo, const float* lse, float* delta, vitk_bhnd dq_f32, vitk_bhnd dk_f32,
                     vitk_bhnd dv_f32, int64_t B, int64_t H, int64_t N, int64_t d, float scale, void* stream);
/* keep[r * cols + c] = 1 if element (r, c) survives dropout(p) under `seed` in the fused kernels, else 0. */
int vitk_dropout_keep(uint8_t* keep, int64_t rows, int64_t cols, float p, uint32_t seed, void* stream);

/* Variable-length (packed) attention -- the NaViT path (na_vit.py:115-169: F.scaled_dot_product_attention with a dense
 * boolean "same image & key not padding" mask, :335-337; and the attention pool :371-387).  Tokens of all images are
 * packed without padding into (T, H*d) matrices addressed as p + n*s_n + h*s_h + d; attention runs per segment
 * (= image): query rows [cu_q[s], cu_q[s+1]) against key rows [cu_k[s], cu_k[s+1]).  No mask is materialised.
 * blk_seg/blk_r0 (int32, device): for every 128-row block of the launch, its segment and its first row inside the
 * segment (query blocks for fwd and dQ, key blocks for dK/dV), SORTED BY SEGMENT (the kernels run blocks that are
 * neighbours in the table on one XCD: blocks of one segment share its K|V rows).  lse, delta: f32 (H, tq_total).
 * 16-bit, d in {32, 48, 64, 80, 96} (ViT-H/14's dim_head 80 runs here as B segments of N rows); nblk * H < 2^31.     */
typedef struct vitk_hnd { void* p; int64_t s_h, s_n; } vitk_hnd;
int vitk_attn_varlen_fwd_bf16(vitk_hnd q, vitk_hnd k, vitk_hnd v, vitk_hnd o, float* lse,
                              const int32_t* cu_q, const int32_t* cu_k, const int32_t* blk_seg, const int32_t* blk_r0,
                              int64_t nblk, int64_t tq_total, int64_t H, int64_t d, float scale, void* stream);
int vitk_attn_varlen_bwd_bf16(vitk_hnd q, vitk_hnd k, vitk_hnd v, vitk_hnd o, vitk_hnd dout, const float* lse, float* delta,
                              vitk_hnd dq, vitk_hnd dk, vitk_hnd dv, const int32_t* cu_q, const int32_t* cu_k,
                              const int32_t* qblk_seg, const int32_t* qblk_r0, int64_t nqblk,
                              const int32_t* kblk_seg, const int32_t* kblk_r0, int64_t nkblk,
                              int64_t tq_total, int64_t H, int64_t d, float scale, void* stream);
/* The same with dropout on the attention matrix (na_vit.py:163 `dropout_p`): keep decision of (head h, packed query row r,
 * key j of the image) = hash(hash(h * tq_total + r ^ seed) + j) >= p * 2^32, as in vitk_attn_fwd_bf16_drop.          */
int vitk_attn_varlen_fwd_bf16_drop(vitk_hnd q, vitk_hnd k, vitk_hnd v, vitk_hnd o, float* lse,
                                   const int32_t* cu_q, const int32_t* cu_k, const int32_t* blk_seg, const int32_t* blk_r0,
                                   int64_t nblk, int64_t tq_total, int64_t H, int64_t d, float scale, float drop_p,
                                   uint32_t drop_seed, void* stream);
int vitk_attn_varlen_bwd_bf16_drop(vitk_hnd q, vitk_hnd k, vitk_hnd v, vitk_hnd o, vitk_hnd dout, const float* lse, float* delta,
                                   vitk_hnd dq, vitk_hnd dk, vitk_hnd dv, const int32_t* cu_q, const int32_t* cu_k,
                                   const int32_t* qblk_seg, const int32_t* qblk_r0, int64_t nqblk,
                                   const int32_t* kblk_seg, const int32_t* kblk_r0, int64_t nkblk,
                                   int64_t tq_total, int64_t H, int64_t d, float scale, float drop_p, uint32_t drop_seed,
                                   void* stream);

/* NaViT q/k normalisation (RMSNorm, na_vit.py:93-101): y = x / max(||x||, 1e-12) * sqrt(d) * gamma[h, :] for every
 * (token, head); x, y viewed (T, H, d) with token strides ldx / ldy (d % 4 == 0, d <= 256: the reference leaves dim_head
 * free, na_vit.py:119; strides % 4 == 0).  rnorm: f32 (T*H) saved for backward.
 * Backward also needs `partials`: f32 workspace of vitk_rmsnorm_heads_rows(T, H) * 64 * ceil(d / 64) floats.    */
int64_t vitk_rmsnorm_heads_rows(int64_t T, int64_t H);
int vitk_rmsnorm_heads_fwd(const void* x, int64_t ldx, const void* gamma, void* y, int64_t ldy, float* rnorm, int dt,
                           int64_t T, int64_t H, int64_t d, void* stream);
int vitk_rmsnorm_heads_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* gamma, const float* rnorm,
                           void* dx, int64_t lddx, void* dgamma, float* partials, int dt,
                           int64_t T, int64_t H, int64_t d, void* stream);

/* Materialising path pieces (nn.Softmax at vit.py:41,59; needed when `attend` has forward hooks,
 * for dim_head != 64 and for f32 validation mode): row softmax of scale*s and its backward.  */
int vitk_softmax_fwd(const void* s, void* p, int dt, int64_t rows, int64_t cols, float scale, void* stream);
/* ds = scale * p * (dp - rowsum(dp * p)) */
int vitk_softmax_bwd(const void* p, const void* dp, void* ds, int dt, int64_t rows, int64_t cols,
                     float scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * Element-wise / data movement
 * ------------------------------------------------------------------------------------------ */
/* Rearrange 'b c (h p1) (w p2) -> b (h w) (p1 p2 c)' (vit.py:100): out[(b*h*w), p1*p2*c]        */
int vitk_patchify(const void* img, void* out, int dt, int64_t B, int64_t C, int64_t H, int64_t W,
                  int64_t p1, int64_t p2, void* stream);
/* Its gradient with respect to the image (autograd through einops.Rearrange, vit.py:100, when the input requires a gradient):
 * dimg (B, C, H, W) <- dpatch (B*h*w, p1*p2*C), every pixel written exactly once.                                           */
int vitk_unpatchify(const void* dpatch, void* dimg, int dt, int64_t B, int64_t C, int64_t H, int64_t W,
                    int64_t p1, int64_t p2, void* stream);
/* Fused first stage of the patch embedding (vit.py:100-101): Rearrange + LayerNorm(patch_dim) with the gather in the load -- y[(b*h*w), 768]
 * = LayerNorm(patch vector) straight from the NCHW image, no `patches` tensor.  Serves 16-bit images of 3 channels with 16 x 16 patches
 * (vitk_patch_ln_serves: 1 / 0); mean / rstd per patch row are kept for the backward.  The backward gives only the parameter gradients
 * (the image needs none): partials = [2][nblk][768] floats (dgamma, dbeta; nblk = vitk_patch_ln_bwd_blocks(B*h*w)), fold them with
 * vitk_colsum_partials.  dy = gradient at the LayerNorm output, (B*h*w, 768), 16-bit.                                             */
int vitk_patch_ln_serves(int dt, int64_t C, int64_t H, int64_t W, int64_t p1, int64_t p2);
int64_t vitk_patch_ln_bwd_blocks(int64_t rows);
int vitk_patch_ln_fwd(const void* img, int dt, const void* w, const void* b, void* y, float* mean, float* rstd, int64_t B,
                      int64_t C, int64_t H, int64_t W, int64_t p1, int64_t p2, float eps, void* stream);
int vitk_patch_ln_bwd_params(const void* dy, const void* img, int dt, const float* mean, const float* rstd, float* partials,
                             int64_t B, int64_t C, int64_t H, int64_t W, int64_t p1, int64_t p2, void* stream);
/* NaViT patch extraction of ONE image, 'c (h p1) (w p2) -> (h w) (c p1 p2)' (na_vit.py:300): out rows
 * [row0, row0 + h*w) of a (T, C*p*p) matrix (leading dimension ld).                                           */
int vitk_patchify_cpp(const void* img, void* out, int dt, int64_t C, int64_t H, int64_t W, int64_t p,
                      int64_t row0, int64_t ld, void* stream);
/* Its adjoint -- the gradient of na_vit.py:300 with respect to the image (the reference is differentiable there): dimg (C, H, W)
 * from rows [row0, row0 + h*w) of dpatch (leading dimension ld); every image element is written exactly once.     */
int vitk_unpatchify_cpp(const void* dpatch, void* dimg, int dt, int64_t C, int64_t H, int64_t W, int64_t p,
                        int64_t row0, int64_t ld, void* stream);
/* out[t, :] = x[t, :] + A[ia[t], :] + B[ib[t], :]   (factorised 2-d positional embedding, na_vit.py:354-359)     */
int vitk_gather_add2(const void* x, const void* A, const int32_t* ia, const void* B, const int32_t* ib, void* out, int dt,
                     int64_t T, int64_t D, void* stream);
/* out[i, :] = sum_{j in [ptr[i], ptr[i+1])} g[rows[j], :]  -- deterministic segmented row sum in CSR form (gradient of
 * the embedding gathers above; also a row gather when every segment has one entry).  g: gdt, out: odt.          */
int vitk_csr_rowsum(const void* g, int gdt, const int32_t* ptr, const int32_t* rows, void* out, int odt,
                    int64_t nseg, int64_t D, void* stream);
/* y = gelu_erf(x) (vit.py:21) ; dx = dy * gelu'(x) */
int vitk_gelu_fwd(const void* x, void* y, int dt, int64_t n, void* stream);
int vitk_gelu_bwd(const void* dy, const void* x, void* dx, int dt, int64_t n, void* stream);
/* out[r, :] = a[r, :] + b[r, :] (+ bias[:]) ; a f32 or T, b T, out f32 or T (residual adds)   */
int vitk_add_rows(const void* a, int adt, const void* b, int bdt, const void* bias, int biasdt,
                  void* out, int odt, int64_t rows, int64_t cols, void* stream);
/* dtype conversion / copy, n elements */
int vitk_cast(const void* x, int xdt, void* y, int ydt, int64_t n, void* stream);
/* The same conversion for `count` tensors (host tables of pointers and element counts) in ceil(count / 64) launches: what
 * torch.autocast does to the weights of a float32 model once per forward (the reference under `torch.autocast`, or accelerate's
 * mixed precision, train_vit_decorr.py:74) and to their gradients on the way back.  float32 tensors 16-byte aligned, 16-bit 8-byte. */
int vitk_cast_many(const void* const* src, void* const* dst, const int64_t* numel, int64_t count, int xdt, int ydt, void* stream);
/* `count` folds of partial rows in ceil(count / 40) launches: dst_j[c] = (flags_j & 1 ? dst_j[c] : 0) + sum_{p < nparts_j} src_j[p * ld_j + c],
 * c < cols_j; dst dtype tag in flags_j >> 4 (VITK_F32 / VITK_BF16 = "T").  What vitk_colsum_partials / vitk_layernorm_bwd_finalize do one
 * job at a time (same order of additions, bit-identical): the bias gradients of nn.Linear (vit.py:20,23,47) and the weight / bias
 * gradients of nn.LayerNorm (vit.py:19,39,69) of a whole transformer layer's backward in one launch.  Host tables, like vitk_cast_many. */
int vitk_fold_many(const float* const* src, void* const* dst, const int64_t* nparts, const int64_t* ld, const int64_t* cols,
                   const int32_t* flags, int64_t count, void* stream);
/* x[b, 0:ncls, :] = cls[0:ncls, :] + pos[0:ncls, :] for every b (vit.py:122-127); x f32 or T   */
int vitk_write_cls_rows(void* x, int xdt, const void* cls, const void* pos, int pdt,
                        int64_t B, int64_t N, int64_t D, int64_t ncls, void* stream);
/* out[b, :] = mean_n x[b, n, :] (vit.py:135 pool='mean') ; dx[b, n, :] = dout[b, :] / N          */
int vitk_mean_pool_fwd(const void* x, int xdt, void* out, int odt, int64_t B, int64_t N, int64_t D, void* stream);
int vitk_mean_pool_bwd(const void* dout, int ddt, void* dx, int xdt, int64_t B, int64_t N, int64_t D, void* stream);
/* Counter-based dropout (nn.Dropout at vit.py:22,24,42,48,109): y = x * keep / (1-p); mask u8. */
int vitk_dropout_fwd(const void* x, void* y, uint8_t* mask, int dt, int64_t n, float p,
                     uint64_t seed, uint64_t offset, void* stream);
int vitk_dropout_bwd(const void* dy, const uint8_t* mask, void* dx, int dt, int64_t n, float p, void* stream);
/* dst[r, c] = c < cols_copy ? src[r, c] : 0 for c < cols_dst  (zero-pads / strips the K columns of a matrix whose inner
 * extent is not a multiple of 32, e.g. patch_dim = 588 of ViT-H/14, so that the MFMA GEMMs can take it)          */
int vitk_copy_cols(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int dt, int64_t rows, int64_t cols_copy,
                   int64_t cols_dst, void* stream);
/* f32 validation mode ON THE MFMA KERNELS: x (f32) is written as six blocks of its three-term bf16 split hi + mid + lo -- block
 * order {hi,hi,mid,hi,lo,mid} (operand_b = 0) or {hi,mid,hi,lo,hi,mid} (operand_b = 1) -- so that ONE 16-bit GEMM over the six-fold
 * reduction extent sums hi.hi + hi.mid + mid.hi + hi.lo + lo.hi + mid.mid = the f32 product to ~2^-22.  Block b lands at
 * out + b * block_stride (+ r * ld_out + c): block_stride = cols, ld_out = 6 * cols concatenates along K (vitk_gemm_nt_bf16);
 * block_stride = rows * ld_out concatenates along M (vitk_gemm_tn_bf16).  bfloat16 library only.                              */
int vitk_split_bf16x3(const float* x, int64_t ldx, void* out, int64_t ld_out, int64_t block_stride, int64_t rows,
                      int64_t cols, int operand_b, void* stream);
/* out[b, i, :] = (i < F ? front[i, :] : x[b, i - F, :]) + (pos ? pos[i, :] : 0) for i < Np + F: torch.cat((tokens, x), dim = 1)
 * + pos[:N] of vit.py:122-127 for all images in one launch.  F < 0: the |F| extra tokens go BEHIND x, out[b, i, :] =
 * (i < Np ? x[b, i, :] : front[i - Np, :]) + pos: pack([x, r]) of simple_vit_with_register_tokens.py:113-115.             */
int vitk_concat_tokens(const void* x, const void* front, const void* pos, void* out, int dt, int64_t B, int64_t Np,
                       int64_t F, int64_t D, void* stream);
/* scatter = 0: dst[b, j, :] = src[b, idx[b, j], :] (x[batch_indices, patch_indices_keep] of vit_with_patch_dropout.py:32);
 * scatter = 1: dst[b, idx[b, j], :] = src[b, j, :] (its adjoint; dst zeroed by the caller).  idx: int32 [B, Kp], distinct per b. */
int vitk_gather_tokens(const void* src, const int32_t* idx, void* dst, int dt, int64_t B, int64_t Np, int64_t Kp, int64_t D,
                       int scatter, void* stream);
/* One Adam (decoupled = 0; torch.optim.Adam as used by train_vit_decorr.py:68-70,110) or AdamW (decoupled = 1) step over a
 * flat range of n elements: param, grad of dtype dt; exp_avg, exp_avg_sq (and the optional f32 master copy of bf16
 * parameters) f32.  step counts from 1 (bias corrections 1 - beta^step are formed on the host in double).  grad is read as
 * grad * grad_scale (loss-scale / gradient-averaging factor).  With the flat gradient buffer of the data-parallel sink
 * and parameters re-homed into one flat buffer this is ONE launch for the whole model.                              */
int vitk_adam_step(void* param, const void* grad, int dt, float* exp_avg, float* exp_avg_sq, float* master, int64_t n,
                   float lr, float beta1, float beta2, float eps, float weight_decay, int decoupled, int64_t step,
                   float grad_scale, void* stream);
/* 2-D transpose out[c][r] = in[r][c] (weights: W^T for the dX GEMMs) */
int vitk_transpose(const void* in, void* out, int dt, int64_t rows, int64_t cols, void* stream);

/* ---- data-parallel gradient exchange -----------------------------------------------------------------
 * What accelerate / DistributedDataParallel do for the reference's training script (train_vit_decorr.py:68-78: all-reduce of
 * every parameter's .grad across ranks), for hosts that bind this library directly: ONE in-place all-reduce of a contiguous
 * gradient range (the flat buffer, or a chunk of it) over RCCL on the caller's stream.  One process per GPU.  Rank 0 calls
 * vitk_comm_unique_id and hands the 128 bytes to the other ranks by its own means (a file, a socket, MPI); every rank then
 * calls vitk_comm_init.  dt: VITK_F32 or the library's 16-bit type; average != 0 divides by the world size (ncclAvg).
 * librccl is opened at first use (no link-time dependency): VITK_E_UNAVAILABLE if it is not there, VITK_E_COMM for RCCL
 * errors.  The Python package itself exchanges gradients through torch.distributed (backend "nccl" = RCCL).              */
#define VITK_COMM_ID_BYTES 128
typedef void* vitk_comm_t;
int vitk_comm_unique_id(void* out128);
int vitk_comm_init(const void* id128, int rank, int world, vitk_comm_t* out);
int vitk_comm_allreduce(vitk_comm_t comm, void* buf, int64_t n, int dt, int average, void* stream);
int vitk_comm_destroy(vitk_comm_t comm);

#ifdef __cplusplus
}
#endif
#endif /* VITK_H */
