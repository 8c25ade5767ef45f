"""Tensor-level wrappers over the C-ABI (one Python function per libvitk entry point).

torch is used here only as the owner of device memory and of the current HIP stream
(``torch.cuda.current_stream().cuda_stream`` is the ``hipStream_t`` on ROCm builds); every
computation happens inside libvitk.so.  All functions enqueue on the CURRENT stream and
return immediately.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib as L
from ._lib import BF16, F32, IDENT, BHND, Mat, RowMap, check

Tensor = torch.Tensor


HALF_DTYPES = (torch.bfloat16, torch.float16)


def dt(t_or_dtype) -> int:
    d = t_or_dtype.dtype if isinstance(t_or_dtype, torch.Tensor) else t_or_dtype
    if d == torch.float32:
        return F32
    if d in HALF_DTYPES:
        return BF16          # "the library's 16-bit type": _lib_for() routes float16 tensors to libvitk_f16.so
    raise L.VitkError(f"unsupported dtype {d}: libvitk computes in float32, bfloat16 or float16")


def _lib_for(*objs):
    """The library instance that serves these operands: libvitk_f16.so as soon as one of them is float16.

    Operands are tensors or the pointer structs built by mat() / bhnd() / hnd() (which remember their tensor's dtype).
    bfloat16 and float16 operands must not meet in one call -- each library knows a single 16-bit type."""
    f16 = b16 = False
    for o in objs:
        d = o.dtype if isinstance(o, torch.Tensor) else getattr(o, "_dtype", None)
        f16 = f16 or d == torch.float16
        b16 = b16 or d == torch.bfloat16
    if f16 and b16:
        raise L.VitkError("float16 and bfloat16 operands in one call: a model must use ONE 16-bit dtype")
    lib = L.load_f16() if f16 else L.load()
    L.note_last_lib(lib)        # check() reads the (thread-local, per-library) error string from the right instance
    return lib


def _p(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def require_device(*ts: Tensor):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.VitkError(
                "vit_pytorch_amd runs on an AMD GPU (HIP) only: got a tensor on "
                f"{t.device}. Move the module and its input to 'cuda'. There is no CPU path.")


# ---- LayerNorm -------------------------------------------------------------------------------
def layernorm_fwd(x: Tensor, w: Tensor, b: Optional[Tensor], y: Tensor, mean: Tensor, rstd: Tensor,
                  rows: int, D: int, eps: float = 1e-5, imap: RowMap = IDENT, omap: RowMap = IDENT,
                  add: Optional[Tensor] = None, add_group: int = 0, add_off: int = 0, y8: Optional[Tensor] = None,
                  scale8: Optional[Tensor] = None, amax64: Optional[Tensor] = None):
    """y8 / scale8 / amax64: optional e4m3 copy of y under the (device) scale, and this call's amax record (fp8.py)."""
    lib = _lib_for(x, w, y, mean, rstd)
    check(lib.vitk_layernorm_fwd_fp8(_p(x), dt(x), _p(w), _p(b), dt(w), _p(y), dt(y), _p(mean), _p(rstd),
                                     rows, D, eps, imap, omap, _p(add), add_group, add_off, _p(y8), _p(scale8), _p(amax64), _stream()),
          "layernorm_fwd")


def layernorm_bwd_blocks(rows: int, D: int) -> int:
    return int(L.load().vitk_layernorm_bwd_blocks(rows, D))


def layernorm_bwd(dy: Tensor, x: Tensor, w: Tensor, mean: Tensor, rstd: Tensor, gin: Optional[Tensor],
                  dx_f32: Optional[Tensor], dx_t: Optional[Tensor], partials: Tensor, colsum_dx: bool,
                  rows: int, D: int, dymap: RowMap = IDENT, xmap: RowMap = IDENT, dxmap: RowMap = IDENT,
                  drop_p: float = 0.0, drop_seed: int = 0):
    lib = _lib_for(dy, x, w, mean, rstd, partials)
    check(lib.vitk_layernorm_bwd_drop(_p(dy), dt(dy), _p(x), dt(x), _p(w), dt(w), _p(mean), _p(rstd), _p(gin),
                                      _p(dx_f32), _p(dx_t), dt(dx_t) if dx_t is not None else F32, _p(partials),
                                      1 if colsum_dx else 0, rows, D, dymap, xmap, dxmap, drop_p, drop_seed & 0xffffffff, _stream()),
          "layernorm_bwd")


def layernorm_bwd_s16(dy: Tensor, x: Tensor, w: Tensor, mean: Tensor, rstd: Tensor, gin: Optional[Tensor], dx_t: Tensor,
                      partials: Tensor, colsum_dx: bool, rows: int, D: int, dymap: RowMap = IDENT, xmap: RowMap = IDENT,
                      dxmap: RowMap = IDENT):
    """LayerNorm backward with the stream gradient in the 16-bit type: dx_t = dx + gin, no float32 output."""
    lib = _lib_for(dy, x, w, mean, rstd, partials, dx_t, gin)
    check(lib.vitk_layernorm_bwd_s16(_p(dy), dt(dy), _p(x), dt(x), _p(w), dt(w), _p(mean), _p(rstd), _p(gin), _p(dx_t), dt(dx_t),
                                     _p(partials), 1 if colsum_dx else 0, rows, D, dymap, xmap, dxmap, _stream()), "layernorm_bwd_s16")


def layernorm_bwd_finalize(partials: Tensor, nblk: int, D: int, dw: Optional[Tensor], db: Optional[Tensor],
                           dcol: Optional[Tensor], odt: int):
    """dcol: float32, or the parameter dtype (then it is written as the bias gradient it is: vitk_layernorm_bwd_finalize_ex)."""
    check(_lib_for(partials, dw, db, dcol).vitk_layernorm_bwd_finalize_ex(_p(partials), nblk, D, _p(dw), _p(db), odt, _p(dcol),
                                                                          F32 if dcol is None else dt(dcol), _stream()),
          "layernorm_bwd_finalize")


def colsum_partials(partials: Tensor, nparts: int, ld: int, cols: int, out: Tensor, accumulate: bool = False):
    check(_lib_for(partials, out).vitk_colsum_partials(_p(partials), nparts, ld, cols, _p(out), dt(out), int(accumulate), _stream()),
          "colsum_partials")


def fold_many(jobs):
    """jobs: [(partials, nparts, ld, cols, out, accumulate)]: out[c] = (accumulate ? out[c] : 0) + sum_p partials[p * ld + c] for every job
    in ceil(len / 40) launches (vitk_fold_many) -- the colsum_partials / layernorm_bwd_finalize of a whole layer's backward at once."""
    import ctypes
    if not jobs:
        return
    n = len(jobs)
    for part, nparts, ld, cols, out, _ in jobs:
        if part.dtype != torch.float32 or not out.is_contiguous() or out.numel() < cols:
            raise L.VitkError("fold_many: float32 partial rows and a contiguous output of at least `cols` elements are required")
    src = (ctypes.c_void_p * n)(*[j[0].data_ptr() for j in jobs])
    dst = (ctypes.c_void_p * n)(*[j[4].data_ptr() for j in jobs])
    npar = (ctypes.c_int64 * n)(*[int(j[1]) for j in jobs])
    lds = (ctypes.c_int64 * n)(*[int(j[2]) for j in jobs])
    cls = (ctypes.c_int64 * n)(*[int(j[3]) for j in jobs])
    flg = (ctypes.c_int32 * n)(*[(1 if j[5] else 0) | (dt(j[4]) << 4) for j in jobs])
    cv = lambda a: ctypes.cast(a, ctypes.c_void_p)
    check(_lib_for(*[j[4] for j in jobs]).vitk_fold_many(cv(src), cv(dst), cv(npar), cv(lds), cv(cls), cv(flg), n, _stream()), "fold_many")


def colsum_ws_floats(rows: int, cols: int) -> int:
    return int(L.load().vitk_colsum_ws_floats(rows, cols))


def colsum(x: Tensor, rows: int, cols: int, ld: int, out: Tensor, ws: Tensor, accumulate: bool = False):
    check(_lib_for(x, out, ws).vitk_colsum(_p(x), dt(x), rows, cols, ld, _p(out), dt(out), int(accumulate), _p(ws), _stream()),
          "colsum")


# ---- GEMMs -----------------------------------------------------------------------------------
def gemm_nt_bf16(A: Tensor, lda: int, W: Tensor, ldw: int, C: Tensor, ldc: int, M: int, N: int, K: int,
                 epilogue: int = L.EPI_NONE, bias: Optional[Tensor] = None, resid: Optional[Tensor] = None,
                 aux: Optional[Tensor] = None):
    check(_lib_for(A, W, C).vitk_gemm_nt_bf16(_p(A), lda, _p(W), ldw, _p(C), ldc, M, N, K, epilogue, _p(bias), _p(resid),
                                     _p(aux), _stream()), "gemm_nt_bf16")


def gemm_nt_bf16_drop(A: Tensor, lda: int, W: Tensor, ldw: int, C: Tensor, ldc: int, M: int, N: int, K: int, epilogue: int,
                      drop_p: float, drop_seed: int, bias: Optional[Tensor] = None, resid: Optional[Tensor] = None,
                      aux: Optional[Tensor] = None, partials: Optional[Tensor] = None):
    """vitk_gemm_nt_bf16 with nn.Dropout(p) fused into the RESID / BIAS_GELU / GELU_BWD epilogue (256-row kernel only)."""
    check(_lib_for(A, W, C).vitk_gemm_nt_bf16_drop(_p(A), lda, _p(W), ldw, _p(C), ldc, M, N, K, epilogue, _p(bias), _p(resid), _p(aux),
                                                   _p(partials), drop_p, drop_seed & 0xffffffff, _stream()), "gemm_nt_bf16_drop")


def _lib_of(dtype):
    """The library flavour that will serve operands of `dtype` (None / anything but float16: the bfloat16 library).  The schedule
    queries below depend on per-library state (vitk_set_cu_reserve), so they must ask the library that will run the GEMM."""
    return L.load_f16() if dtype == torch.float16 else L.load()


def gemm_nt_colsum_rows(M: int, N: int, K: int, ldc: int, dtype=None) -> int:
    return int(_lib_of(dtype).vitk_gemm_nt_colsum_rows(M, N, K, ldc))


def gemm_nt_plan(M: int, N: int, K: int, ldc: int) -> dict:
    """Tile schedule of the persistent NT kernel (vitk_gemm_nt_plan)."""
    import ctypes
    out = (ctypes.c_int32 * 5)()
    check(L.load().vitk_gemm_nt_plan(M, N, K, ldc, ctypes.cast(out, ctypes.c_void_p)), "gemm_nt_plan")
    return {"persistent": bool(out[0]), "tiles_m256": out[1], "tiles_m128": out[2], "workgroups": out[3], "tiles_n": out[4]}


def pack_w_nt_bytes(rows: int, reduction: int) -> int:
    return int(L.load().vitk_pack_w_nt_bytes(rows, reduction))


def pack_w_nt(W: Tensor, ldw: int, N: int, Kd: int, out: Optional[Tensor], out_t: Optional[Tensor]):
    """K-blocked copies of a weight (see vitk.h): `out` for y = x W^T, `out_t` for dX = dY W; pass them as W with ldw = 0."""
    check(_lib_for(W, out, out_t).vitk_pack_w_nt(_p(W), ldw, N, Kd, _p(out), _p(out_t), _stream()), "pack_w_nt")


def pack_w_nt_many(rows):
    """rows: list of (W, N, Kd, out or None, out_t or None) -- vitk_pack_w_nt for all of them in ceil(jobs / 96) launches."""
    import ctypes
    rows = [r for r in rows if r[3] is not None or r[4] is not None]
    if not rows:
        return
    n = len(rows)
    pw = (ctypes.c_void_p * n)(*[r[0].data_ptr() for r in rows])
    ld = (ctypes.c_int64 * n)(*[r[2] for r in rows])
    nn = (ctypes.c_int64 * n)(*[r[1] for r in rows])
    kk = (ctypes.c_int64 * n)(*[r[2] for r in rows])
    po = (ctypes.c_void_p * n)(*[(r[3].data_ptr() if r[3] is not None else None) for r in rows])
    pt = (ctypes.c_void_p * n)(*[(r[4].data_ptr() if r[4] is not None else None) for r in rows])
    c = lambda a: ctypes.cast(a, ctypes.c_void_p)
    check(_lib_for(*[r[0] for r in rows]).vitk_pack_w_nt_many(c(pw), c(ld), c(nn), c(kk), c(po), c(pt), n, _stream()), "pack_w_nt_many")


def gemm_nt_bf16_gelu_bwd_colsum(A: Tensor, lda: int, W: Tensor, ldw: int, C: Tensor, ldc: int, M: int, N: int, K: int,
                                 aux: Tensor, partials: Tensor):
    check(_lib_for(A, W, C, aux, partials).vitk_gemm_nt_bf16_gelu_bwd_colsum(_p(A), lda, _p(W), ldw, _p(C), ldc, M, N, K, _p(aux), _p(partials),
                                                     _stream()), "gemm_nt_bf16_gelu_bwd_colsum")


def gemm_nt_bf16_mul_aux_colsum(A: Tensor, lda: int, W: Tensor, ldw: int, C: Tensor, ldc: int, M: int, N: int, K: int,
                                aux: Tensor, partials: Optional[Tensor]):
    """C = (A . W^T) * aux (EPI_MUL_AUX: aux = the gelu' factor an EPI_BIAS_GELU_DG forward stored) + column sums of C."""
    check(_lib_for(A, W, C, aux).vitk_gemm_nt_bf16_mul_aux_colsum(_p(A), lda, _p(W), ldw, _p(C), ldc, M, N, K, _p(aux), _p(partials),
                                                     _stream()), "gemm_nt_bf16_mul_aux_colsum")


def gemm_nt_bf16_mul_aux8_colsum(A: Tensor, lda: int, W: Tensor, ldw: int, C: Tensor, ldc: int, M: int, N: int, K: int,
                                 aux8: Tensor, partials: Optional[Tensor]):
    """C = (A . W^T) * 0.005 (aux8 - 27) (EPI_MUL_AUX8: aux8 = the 8-bit codes of the gelu' factor an EPI_BIAS_GELU_DG8 forward stored) + column
    sums of C."""
    if aux8.dtype != torch.uint8:
        raise L.VitkError("gemm_nt_bf16_mul_aux8_colsum: aux8 must be uint8")
    check(_lib_for(A, W, C).vitk_gemm_nt_bf16_mul_aux8_colsum(_p(A), lda, _p(W), ldw, _p(C), ldc, M, N, K, _p(aux8), _p(partials),
                                                              _stream()), "gemm_nt_bf16_mul_aux8_colsum")


def set_cu_reserve(cus: int, dtype=None):
    """CUs the weight-gradient GEMMs leave to other kernels (vitk_set_cu_reserve; process-wide per library flavour)."""
    lib = L.load_f16() if dtype == torch.float16 else L.load()
    check(lib.vitk_set_cu_reserve(int(cus)), "set_cu_reserve")


def gemm_tn_splits(M: int, N: int, K: int, dtype=None) -> int:
    return int(_lib_of(dtype).vitk_gemm_tn_splits(M, N, K))


def gemm_tn_bf16(dY: Tensor, ldy: int, X: Tensor, ldx: int, dW: Tensor, ldo: int, M: int, N: int, K: int,
                 ws: Tensor, splits: int, accumulate: bool = False):
    check(_lib_for(dY, X, dW, ws).vitk_gemm_tn_bf16(_p(dY), ldy, _p(X), ldx, _p(dW), dt(dW), ldo, int(accumulate), M, N, K,
                                     _p(ws), splits, _stream()), "gemm_tn_bf16")


def gemm_tn_pair_splits(M: int, N0: int, K0: int, N1: int, K1: int, dtype=None) -> int:
    """Split count of the paired weight-gradient launch, from the library that will run it: the count depends on that library's
    vitk_set_cu_reserve() and vitk_gemm_tn_bf16_pair insists on its own answer."""
    return int(_lib_of(dtype).vitk_gemm_tn_pair_splits(M, N0, K0, N1, K1))


def gemm_tn_bf16_pair(dY0: Tensor, ldy0: int, X0: Tensor, ldx0: int, dW0: Tensor, dY1: Tensor, ldy1: int, X1: Tensor, ldx1: int, dW1: Tensor,
                      M: int, ws: Tensor, splits: int, accumulate0: bool = False, accumulate1: bool = False):
    """dW0 (N0, K0) = dY0^T X0 and dW1 (N1, K1) = dY1^T X1 over the same M token rows in one launch (vitk_gemm_tn_bf16_pair)."""
    (N0, K0), (N1, K1) = dW0.shape, dW1.shape
    check(_lib_for(dY0, X0, dW0, dY1, X1, dW1, ws).vitk_gemm_tn_bf16_pair(_p(dY0), ldy0, _p(X0), ldx0, _p(dW0), K0, int(accumulate0), N0, K0,
                                                                    _p(dY1), ldy1, _p(X1), ldx1, _p(dW1), K1, int(accumulate1), N1, K1,
                                                                    dt(dW0), M, _p(ws), splits, _stream()), "gemm_tn_bf16_pair")


def mat(t: Tensor, s_row: int, s_col: int, s_b1: int = 0, s_b2: int = 0, offset: int = 0) -> Mat:
    m = Mat(t.data_ptr() + offset * t.element_size(), dt(t), s_b1, s_b2, s_row, s_col)
    m._dtype = t.dtype
    return m


def gemm_generic(A: Mat, B: Mat, Cm: Mat, M: int, N: int, K: int, nb1: int = 1, nb2: int = 1,
                 bias: Optional[Tensor] = None, alpha: float = 1.0, beta: float = 0.0):
    check(_lib_for(A, B, Cm).vitk_gemm_generic(A, B, Cm, _p(bias), dt(bias) if bias is not None else F32, nb1, nb2, M, N, K,
                                     alpha, beta, _stream()), "gemm_generic")


# ---- attention -------------------------------------------------------------------------------
def bhnd(t: Tensor, s_b: int, s_h: int, s_n: int, offset: int = 0) -> BHND:
    r = BHND(t.data_ptr() + offset * t.element_size(), s_b, s_h, s_n)
    r._dtype = t.dtype
    return r


def attn_fwd_bf16(q: BHND, k: BHND, v: BHND, o: BHND, lse: Tensor, B: int, H: int, N: int, d: int, scale: float,
                  drop_p: float = 0.0, drop_seed: int = 0):
    check(_lib_for(q, k, v, o, lse).vitk_attn_fwd_bf16_drop(q, k, v, o, _p(lse), B, H, N, d, scale, drop_p, drop_seed & 0xffffffff,
                                                           _stream()), "attn_fwd_bf16")


def attn_bwd_bf16(q: BHND, k: BHND, v: BHND, o: BHND, dout: BHND, lse: Tensor, delta: Tensor, dq: BHND, dk: BHND,
                  dv: BHND, B: int, H: int, N: int, d: int, scale: float, drop_p: float = 0.0, drop_seed: int = 0):
    check(_lib_for(q, k, v, o, dout, lse, delta, dq, dk, dv).vitk_attn_bwd_bf16_drop(q, k, v, o, dout, _p(lse), _p(delta), dq, dk, dv, B, H, N, d, scale,
                                                                                    drop_p, drop_seed & 0xffffffff, _stream()),
          "attn_bwd_bf16")


def split2(x: Tensor, hi: Tensor, lo: Tensor):
    """f32 tensor -> hi + lo 16-bit terms (operands of the f32-accurate attention flavour)."""
    check(_lib_for(hi, lo).vitk_split2(_p(x), _p(hi), _p(lo), x.numel(), _stream()), "split2")


def attn_fwd_x2(q: Tuple[BHND, BHND], k: Tuple[BHND, BHND], v: Tuple[BHND, BHND], o: BHND, lse: Tensor, B: int, H: int, N: int, d: int,
                scale: float):
    check(_lib_for(q[0], k[0], v[0]).vitk_attn_fwd_x2(q[0], q[1], k[0], k[1], v[0], v[1], o, _p(lse), B, H, N, d, scale, _stream()), "attn_fwd_x2")


def attn_bwd_x2(q: Tuple[BHND, BHND], k: Tuple[BHND, BHND], v: Tuple[BHND, BHND], o: BHND, dout: Tuple[BHND, BHND], lse: Tensor,
                delta: Tensor, dq: BHND, dk: BHND, dv: BHND, B: int, H: int, N: int, d: int, scale: float):
    check(_lib_for(q[0], k[0], v[0]).vitk_attn_bwd_x2(q[0], q[1], k[0], k[1], v[0], v[1], o, dout[0], dout[1], _p(lse), _p(delta), dq, dk, dv,
                                                      B, H, N, d, scale, _stream()), "attn_bwd_x2")


def softmax_fwd(s: Tensor, p: Tensor, rows: int, cols: int, scale: float):
    check(_lib_for(s, p).vitk_softmax_fwd(_p(s), _p(p), dt(s), rows, cols, scale, _stream()), "softmax_fwd")


def softmax_bwd(p: Tensor, dp: Tensor, ds: Tensor, rows: int, cols: int, scale: float):
    check(_lib_for(p, dp, ds).vitk_softmax_bwd(_p(p), _p(dp), _p(ds), dt(p), rows, cols, scale, _stream()), "softmax_bwd")


# ---- fused patch gather + LayerNorm(patch_dim) (vit.py:100-101) ------------------------------------
def patch_ln_serves(img: Tensor, C: int, H: int, W: int, p1: int, p2: int) -> bool:
    return bool(_lib_for(img).vitk_patch_ln_serves(dt(img), C, H, W, p1, p2))


def patch_ln_bwd_blocks(rows: int) -> int:
    return int(L.load().vitk_patch_ln_bwd_blocks(rows))


def patch_ln_fwd(img: Tensor, w: Tensor, b: Optional[Tensor], y: Tensor, mean: Tensor, rstd: Tensor, B: int, C: int, H: int, W: int,
                 p1: int, p2: int, eps: float = 1e-5):
    check(_lib_for(img, w, y).vitk_patch_ln_fwd(_p(img), dt(img), _p(w), _p(b), _p(y), _p(mean), _p(rstd), B, C, H, W, p1, p2, eps, _stream()),
          "patch_ln_fwd")


def patch_ln_bwd_params(dy: Tensor, img: Tensor, mean: Tensor, rstd: Tensor, partials: Tensor, B: int, C: int, H: int, W: int, p1: int, p2: int):
    check(_lib_for(dy, img).vitk_patch_ln_bwd_params(_p(dy), _p(img), dt(img), _p(mean), _p(rstd), _p(partials), B, C, H, W, p1, p2, _stream()),
          "patch_ln_bwd_params")


# ---- element-wise ------------------------------------------------------------------------------
def patchify(img: Tensor, out: Tensor, B: int, C: int, H: int, W: int, p1: int, p2: int):
    check(_lib_for(img, out).vitk_patchify(_p(img), _p(out), dt(img), B, C, H, W, p1, p2, _stream()), "patchify")


def unpatchify(dpatch: Tensor, dimg: Tensor, B: int, C: int, H: int, W: int, p1: int, p2: int):
    """gradient of patchify with respect to the image: (B*h*w, p1*p2*C) -> (B, C, H, W)"""
    check(_lib_for(dpatch, dimg).vitk_unpatchify(_p(dpatch), _p(dimg), dt(dpatch), B, C, H, W, p1, p2, _stream()), "unpatchify")


def gelu_fwd(x: Tensor, y: Tensor):
    check(_lib_for(x, y).vitk_gelu_fwd(_p(x), _p(y), dt(x), x.numel(), _stream()), "gelu_fwd")


def gelu_bwd(dy: Tensor, x: Tensor, dx: Tensor):
    check(_lib_for(dy, x, dx).vitk_gelu_bwd(_p(dy), _p(x), _p(dx), dt(x), x.numel(), _stream()), "gelu_bwd")


def add_rows(a: Tensor, b: Tensor, bias: Optional[Tensor], out: Tensor, rows: int, cols: int):
    check(_lib_for(a, b, out).vitk_add_rows(_p(a), dt(a), _p(b), dt(b), _p(bias), dt(bias) if bias is not None else dt(b),
                                 _p(out), dt(out), rows, cols, _stream()), "add_rows")


def cast(x: Tensor, y: Tensor):
    check(_lib_for(x, y).vitk_cast(_p(x), dt(x), _p(y), dt(y), x.numel(), _stream()), "cast")


def cast_many(srcs, dsts):
    """dsts[i] <- srcs[i] (dtype conversion) for lists of contiguous tensors, one dtype per side: ceil(len / 64) launches."""
    import ctypes
    if not srcs:
        return
    n = len(srcs)
    if len(dsts) != n or any(s.numel() != d.numel() or not s.is_contiguous() or not d.is_contiguous() for s, d in zip(srcs, dsts)):
        raise L.VitkError("cast_many: lists of contiguous tensors with matching element counts are required")
    if len({s.dtype for s in srcs}) != 1 or len({d.dtype for d in dsts}) != 1:
        raise L.VitkError("cast_many: one dtype per side")
    ps = (ctypes.c_void_p * n)(*[s.data_ptr() for s in srcs])
    pd = (ctypes.c_void_p * n)(*[d.data_ptr() for d in dsts])
    ne = (ctypes.c_int64 * n)(*[s.numel() for s in srcs])
    check(_lib_for(srcs[0], dsts[0]).vitk_cast_many(ctypes.cast(ps, ctypes.c_void_p), ctypes.cast(pd, ctypes.c_void_p), ctypes.cast(ne, ctypes.c_void_p),
                                                  n, dt(srcs[0]), dt(dsts[0]), _stream()), "cast_many")


def write_cls_rows(x: Tensor, cls: Tensor, pos: Tensor, B: int, N: int, D: int, ncls: int):
    check(_lib_for(x, cls, pos).vitk_write_cls_rows(_p(x), dt(x), _p(cls), _p(pos), dt(pos), B, N, D, ncls, _stream()),
          "write_cls_rows")


def mean_pool_fwd(x: Tensor, out: Tensor, B: int, N: int, D: int):
    check(_lib_for(x, out).vitk_mean_pool_fwd(_p(x), dt(x), _p(out), dt(out), B, N, D, _stream()), "mean_pool_fwd")


def mean_pool_bwd(dout: Tensor, dx: Tensor, B: int, N: int, D: int):
    check(_lib_for(dout, dx).vitk_mean_pool_bwd(_p(dout), dt(dout), _p(dx), dt(dx), B, N, D, _stream()), "mean_pool_bwd")


def dropout_fwd(x: Tensor, y: Tensor, mask: Tensor, p: float, seed: int, offset: int):
    check(_lib_for(x, y, mask).vitk_dropout_fwd(_p(x), _p(y), _p(mask), dt(x), x.numel(), p, seed, offset, _stream()), "dropout_fwd")


def dropout_bwd(dy: Tensor, mask: Tensor, dx: Tensor, p: float):
    check(_lib_for(dy, mask, dx).vitk_dropout_bwd(_p(dy), _p(mask), _p(dx), dt(dy), dy.numel(), p, _stream()), "dropout_bwd")


def transpose(x: Tensor, out: Tensor, rows: int, cols: int):
    check(_lib_for(x, out).vitk_transpose(_p(x), _p(out), dt(x), rows, cols, _stream()), "transpose")


# ---- NaViT: packed / variable-length path ----------------------------------------------------------
def hnd(t: Tensor, s_h: int, s_n: int, offset: int = 0) -> L.HND:
    r = L.HND(t.data_ptr() + offset * t.element_size(), s_h, s_n)
    r._dtype = t.dtype
    return r


def attn_varlen_fwd_bf16(q: L.HND, k: L.HND, v: L.HND, o: L.HND, lse: Tensor, cu_q: Tensor, cu_k: Tensor, blk_seg: Tensor,
                         blk_r0: Tensor, nblk: int, tq_total: int, H: int, d: int, scale: float, drop_p: float = 0.0, drop_seed: int = 0):
    check(_lib_for(q, k, v, o, lse, cu_q, cu_k, blk_seg, blk_r0).vitk_attn_varlen_fwd_bf16_drop(q, k, v, o, _p(lse), _p(cu_q), _p(cu_k), _p(blk_seg), _p(blk_r0), nblk,
                                             tq_total, H, d, scale, drop_p, drop_seed & 0xffffffff, _stream()), "attn_varlen_fwd_bf16")


def attn_varlen_bwd_bf16(q: L.HND, k: L.HND, v: L.HND, o: L.HND, dout: L.HND, lse: Tensor, delta: Tensor, dq: L.HND,
                         dk: L.HND, dv: L.HND, cu_q: Tensor, cu_k: Tensor, qblk_seg: Tensor, qblk_r0: Tensor, nqblk: int,
                         kblk_seg: Tensor, kblk_r0: Tensor, nkblk: int, tq_total: int, H: int, d: int, scale: float,
                         drop_p: float = 0.0, drop_seed: int = 0):
    check(_lib_for(q, k, v, o, dout, lse, delta, dq, dk, dv, cu_q, cu_k, qblk_seg, qblk_r0, kblk_seg, kblk_r0).vitk_attn_varlen_bwd_bf16_drop(q, k, v, o, dout, _p(lse), _p(delta), dq, dk, dv, _p(cu_q), _p(cu_k),
                                             _p(qblk_seg), _p(qblk_r0), nqblk, _p(kblk_seg), _p(kblk_r0), nkblk,
                                             tq_total, H, d, scale, drop_p, drop_seed & 0xffffffff, _stream()), "attn_varlen_bwd_bf16")


def rmsnorm_heads_rows(T: int, H: int) -> int:
    return int(L.load().vitk_rmsnorm_heads_rows(T, H))


def rmsnorm_heads_partials(T: int, H: int, d: int) -> int:
    """floats of the `partials` workspace of rmsnorm_heads_bwd: vitk_rmsnorm_heads_rows(T, H) rows of 64 * ceil(d / 64)"""
    return rmsnorm_heads_rows(T, H) * ((d + 63) // 64 * 64)


def rmsnorm_heads_fwd(x: Tensor, ldx: int, gamma: Tensor, y: Tensor, ldy: int, rnorm: Tensor, T: int, H: int, d: int,
                      x_off: int = 0):
    check(_lib_for(x, gamma, y, rnorm).vitk_rmsnorm_heads_fwd(x.data_ptr() + x_off * x.element_size(), ldx, _p(gamma), _p(y), ldy, _p(rnorm),
                                          dt(x), T, H, d, _stream()), "rmsnorm_heads_fwd")


def rmsnorm_heads_bwd(dy: Tensor, lddy: int, x: Tensor, ldx: int, gamma: Tensor, rnorm: Tensor, dx: Tensor, lddx: int,
                      dgamma: Tensor, partials: Tensor, T: int, H: int, d: int, x_off: int = 0, dx_off: int = 0):
    check(_lib_for(dy, x, gamma, rnorm, dx, dgamma, partials).vitk_rmsnorm_heads_bwd(_p(dy), lddy, x.data_ptr() + x_off * x.element_size(), ldx, _p(gamma), _p(rnorm),
                                          dx.data_ptr() + dx_off * dx.element_size(), lddx, _p(dgamma), _p(partials),
                                          dt(x), T, H, d, _stream()), "rmsnorm_heads_bwd")


def patchify_cpp(img: Tensor, out: Tensor, C: int, H: int, W: int, p: int, row0: int, ld: int):
    check(_lib_for(img, out).vitk_patchify_cpp(_p(img), _p(out), dt(img), C, H, W, p, row0, ld, _stream()), "patchify_cpp")


def unpatchify_cpp(dpatch: Tensor, dimg: Tensor, C: int, H: int, W: int, p: int, row0: int, ld: int):
    check(_lib_for(dpatch, dimg).vitk_unpatchify_cpp(_p(dpatch), _p(dimg), dt(dpatch), C, H, W, p, row0, ld, _stream()), "unpatchify_cpp")


def gather_add2(x: Tensor, A: Tensor, ia: Tensor, B: Tensor, ib: Tensor, out: Tensor, T: int, D: int):
    check(_lib_for(x, A, ia, B, ib, out).vitk_gather_add2(_p(x), _p(A), _p(ia), _p(B), _p(ib), _p(out), dt(x), T, D, _stream()), "gather_add2")


def csr_rowsum(g: Tensor, ptr: Tensor, rows: Tensor, out: Tensor, nseg: int, D: int):
    check(_lib_for(g, ptr, rows, out).vitk_csr_rowsum(_p(g), dt(g), _p(ptr), _p(rows), _p(out), dt(out), nseg, D, _stream()), "csr_rowsum")


def copy_cols(src: Tensor, ld_src: int, dst: Tensor, ld_dst: int, rows: int, cols_copy: int, cols_dst: int):
    check(_lib_for(src, dst).vitk_copy_cols(_p(src), ld_src, _p(dst), ld_dst, dt(src), rows, cols_copy, cols_dst, _stream()), "copy_cols")


def split_bf16x3(x: Tensor, ldx: int, out: Tensor, ld_out: int, block_stride: int, rows: int, cols: int, operand_b: bool):
    """f32 -> six blocks of its hi / mid / lo bfloat16 split (vitk_split_bf16x3); always the bfloat16 library."""
    check(L.load().vitk_split_bf16x3(_p(x), ldx, _p(out), ld_out, block_stride, rows, cols, int(operand_b), _stream()), "split_bf16x3")


def concat_tokens(x: Tensor, front: Optional[Tensor], pos: Optional[Tensor], out: Tensor, B: int, Np: int, F: int, D: int):
    check(_lib_for(x, front, pos, out).vitk_concat_tokens(_p(x), _p(front), _p(pos), _p(out), dt(x), B, Np, F, D, _stream()), "concat_tokens")


def gather_tokens(src: Tensor, idx: Tensor, dst: Tensor, B: int, Np: int, Kp: int, D: int, scatter: bool = False):
    check(_lib_for(src, dst).vitk_gather_tokens(_p(src), _p(idx), _p(dst), dt(src), B, Np, Kp, D, int(scatter), _stream()), "gather_tokens")


def adam_step(param: Tensor, grad: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, master: Optional[Tensor], n: int, lr: float,
              beta1: float, beta2: float, eps: float, weight_decay: float, decoupled: bool, step: int, grad_scale: float = 1.0):
    check(_lib_for(param, grad, exp_avg, exp_avg_sq).vitk_adam_step(_p(param), _p(grad), dt(param), _p(exp_avg), _p(exp_avg_sq), _p(master), n, lr, beta1, beta2,
                                  eps, weight_decay, int(decoupled), step, grad_scale, _stream()), "adam_step")


# ---- fp8 (e4m3) operands ---------------------------------------------------------------------------------------------
def gemm_nt_fp8(A8: Tensor, lda: int, W8: Tensor, ldw: int, C: Tensor, ldc: int, M: int, N: int, K: int, alpha: float,
                epilogue: int = L.EPI_NONE, bias: Optional[Tensor] = None, resid: Optional[Tensor] = None,
                aux: Optional[Tensor] = None):
    """A8, W8: uint8 / float8_e4m3fn storage of e4m3 values; C, bias, aux: the 16-bit model dtype (selects the library)."""
    check(_lib_for(C, bias, aux).vitk_gemm_nt_fp8(_p(A8), lda, _p(W8), ldw, _p(C), ldc, M, N, K, epilogue, _p(bias), _p(resid),
                                                 _p(aux), alpha, _stream()), "gemm_nt_fp8")


def fp8_amax_scale(x: Tensor, scale2: Tensor):
    check(_lib_for(x).vitk_fp8_amax_scale(_p(x), dt(x), x.numel(), _p(scale2), _stream()), "fp8_amax_scale")


def quantize_fp8(x: Tensor, out: Tensor, scale_dev: Optional[Tensor] = None, scale: float = 1.0):
    check(_lib_for(x).vitk_quantize_fp8(_p(x), dt(x), _p(out), x.numel(), _p(scale_dev), scale, _stream()), "quantize_fp8")


def dropout_keep(keep: Tensor, rows: int, cols: int, p: float, seed: int):
    """uint8 (rows, cols): the keep decisions the fused dropout kernels take for (p, seed) -- a test hook."""
    check(L.load().vitk_dropout_keep(_p(keep), rows, cols, p, seed & 0xffffffff, _stream()), "dropout_keep")


def gemm_nt_fp8_ex(A: Tensor, lda: int, W: Tensor, ldw: int, C: Tensor, ldc: int, M: int, N: int, K: int, epilogue: int, *,
                   a_is_fp8: bool, bias: Optional[Tensor] = None, resid: Optional[Tensor] = None, aux: Optional[Tensor] = None,
                   alpha: float = 1.0, alpha_a: Optional[Tensor] = None, alpha_w: Optional[Tensor] = None,
                   c8: Optional[Tensor] = None, c8_scale: Optional[Tensor] = None, c8_amax64: Optional[Tensor] = None):
    """The NT GEMM with e4m3 or 16-bit operands, device-resident inverse scales and (BIAS_GELU) an e4m3 copy of the output."""
    check(_lib_for(C, bias, aux, None if a_is_fp8 else A).vitk_gemm_nt_fp8_ex(
        _p(A), lda, int(a_is_fp8), _p(W), ldw, _p(C), ldc, M, N, K, epilogue, _p(bias), _p(resid), _p(aux), alpha, _p(alpha_a),
        _p(alpha_w), _p(c8), _p(c8_scale), _p(c8_amax64), _stream()), "gemm_nt_fp8_ex")


def fp8_update_scales(amax64: Tensor, scales2: Tensor, nslots: int):
    check(L.load().vitk_fp8_update_scales(_p(amax64), _p(scales2), nslots, _stream()), "fp8_update_scales")


# ---- fp8 for the backward / out-projection (fp8.py, round 3) ---------------------------------------------------------------
A_16BIT, A_E4M3, A_E5M2 = 0, 1, 2       # a_kind of gemm_nt_fp8_v2
FMT_E4M3, FMT_E5M2 = 0, 1               # fmt of quantize_fp8_delayed


def gemm_nt_fp8_v2(A: Tensor, lda: int, W: Tensor, ldw: int, C: Tensor, ldc: int, M: int, N: int, K: int, epilogue: int, *,
                   a_kind: int, bias: Optional[Tensor] = None, resid: Optional[Tensor] = None, aux: Optional[Tensor] = None,
                   partials: Optional[Tensor] = None, alpha: float = 1.0, alpha_a: Optional[Tensor] = None,
                   alpha_w: Optional[Tensor] = None, c8: Optional[Tensor] = None, c8_scale: Optional[Tensor] = None,
                   c8_amax64: Optional[Tensor] = None, k128: bool = False):
    """vitk_gemm_nt_fp8_v2: e4m3 / e5m2 / 16-bit A against an e4m3 W, every epilogue of the 16-bit GEMM (e5m2: NONE, GELU_BWD),
    optional bias-gradient column sums (GELU_BWD), optional K = 128 MFMA."""
    check(_lib_for(C, bias, aux, A if a_kind == A_16BIT else None).vitk_gemm_nt_fp8_v2(
        _p(A), lda, int(a_kind), _p(W), ldw, _p(C), ldc, M, N, K, epilogue, _p(bias), _p(resid), _p(aux), _p(partials), alpha,
        _p(alpha_a), _p(alpha_w), _p(c8), _p(c8_scale), _p(c8_amax64), 1 if k128 else 0, _stream()), "gemm_nt_fp8_v2")


def gemm_nt_fp8_colsum_rows(M: int, N: int, K: int, ldc: int) -> int:
    return int(L.load().vitk_gemm_nt_fp8_colsum_rows(M, N, K, ldc))


def quantize_fp8_delayed(x: Tensor, out8: Optional[Tensor], scale2: Optional[Tensor], amax64: Optional[Tensor], fmt: int):
    """One pass over x: fp8 copy under the PREVIOUS step's scale (out8 None: none) + this step's amax record (amax64 None: none)."""
    check(_lib_for(x).vitk_quantize_fp8_delayed(_p(x), dt(x), _p(out8), x.numel(), _p(scale2), _p(amax64), int(fmt), _stream()),
          "quantize_fp8_delayed")


def fp8_update_scales_fmt(amax64: Tensor, scales2: Tensor, nslots: int, fmax: Optional[Tensor]):
    check(L.load().vitk_fp8_update_scales_fmt(_p(amax64), _p(scales2), nslots, _p(fmax), _stream()), "fp8_update_scales_fmt")


def gemm_tn_fp8_splits(M: int, N: int, K: int, k128: bool = False) -> int:
    return int(L.load().vitk_gemm_tn_fp8_splits(M, N, K, 1 if k128 else 0))


def gemm_tn_fp8(dY8: Tensor, ldy: int, X8: Tensor, ldx: int, dW: Tensor, ldo: int, M: int, N: int, K: int, ws: Tensor, splits: int, *,
                alpha_y: Optional[Tensor] = None, alpha_x: Optional[Tensor] = None, k128: bool = False, accumulate: bool = False):
    """dW (N, K) = alpha_y alpha_x dY8^T X8 with e5m2 gradients and e4m3 activations (uint8 storage); dW in the model dtype."""
    check(_lib_for(dW).vitk_gemm_tn_fp8(_p(dY8), ldy, _p(X8), ldx, _p(dW), dt(dW), ldo, int(accumulate), M, N, K, _p(ws), splits,
                                        _p(alpha_y), _p(alpha_x), 1 if k128 else 0, _stream()), "gemm_tn_fp8")
