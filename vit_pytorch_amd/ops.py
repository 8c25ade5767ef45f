"""Host-side op layer: shape logic + kernel selection on top of the C-ABI wrappers.

Every function here ends in libvitk kernels (the 16-bit MFMA kernels when the extents allow,
otherwise the generic coverage kernels).  No torch compute ops are used on the data path;
torch only allocates buffers (``torch.empty``) and carries the stream.

dtype policy
  * model dtype T = dtype of the parameters: torch.bfloat16 (production), torch.float16 (model.half(): the same
    kernels from libvitk_f16.so) or torch.float32 (validation mode: same host logic, f32 kernels).
  * the residual stream and its gradient are always float32 (also in the 16-bit modes); everything
    that feeds a GEMM is T.
  * `drop=(p, seed)` arguments: nn.Dropout fused into the kernel that produces the tensor (engine.TransformerFn).
"""
from __future__ import annotations

import os
import threading
import weakref
from typing import Optional, Tuple

import torch

from . import _lib as L
from . import kernels as K
from ._lib import IDENT, RowMap
from ._epoch import caller_grad_mode, weight_key
from .segments import uniform_segments

Tensor = torch.Tensor
BF16 = torch.bfloat16
HALF = (torch.bfloat16, torch.float16)      # the two 16-bit model dtypes (libvitk.so / libvitk_f16.so)
F32 = torch.float32
LN_EPS = 1e-5


def empty(shape, dtype, like: Tensor) -> Tensor:
    return torch.empty(shape, dtype=dtype, device=like.device)


# ---- LayerNorm ---------------------------------------------------------------------------------
def ln_fwd(x: Tensor, w: Tensor, b: Optional[Tensor], rows: int, D: int, out: Tensor,
           imap: RowMap = IDENT, omap: RowMap = IDENT, add: Optional[Tensor] = None,
           add_group: int = 0, add_off: int = 0, f8=None) -> Tuple[Tensor, Tensor]:
    """nn.LayerNorm(D) over `rows` logical rows (vit.py:19,39,69,101,103). Returns (mean, rstd).
    f8 = (y8 or None, scale tensor, amax words): e4m3 copy of the output / amax record for the fp8 forward (fp8.py)."""
    mean = empty((rows,), F32, x)
    rstd = empty((rows,), F32, x)
    y8, sc, am = f8 if f8 is not None else (None, None, None)
    K.layernorm_fwd(x, w, b, out, mean, rstd, rows, D, LN_EPS, imap, omap, add, add_group, add_off, y8, sc, am)
    return mean, rstd


def ln_bwd(dy: Tensor, x: Tensor, w: Tensor, mean: Tensor, rstd: Tensor, rows: int, D: int, *,
           gin: Optional[Tensor] = None, dx_f32: Optional[Tensor] = None, dx_t: Optional[Tensor] = None,
           dw: Optional[Tensor] = None, db: Optional[Tensor] = None, dcol: Optional[Tensor] = None,
           dymap: RowMap = IDENT, xmap: RowMap = IDENT, dxmap: RowMap = IDENT, drop: Optional[Tuple[float, int]] = None):
    """LayerNorm backward; dw/db are (D,) outputs of dtype T, dcol (D,) float32 or T = column sums of dx (the bias gradient of
    the Linear in front of the norm)."""
    nblk = K.layernorm_bwd_blocks(rows, D)
    nslab = 3 if dcol is not None else 2
    partials = empty((nslab * nblk * D,), F32, x)
    if gin is not None and gin.dtype in HALF:          # 16-bit gradient stream (grad_stream_16): dx_t = dx + gin is the new stream
        if dx_f32 is not None or dx_t is None or dx_t.dtype != gin.dtype or drop or D % 4:
            raise L.VitkError("ln_bwd: a 16-bit stream gradient comes with a 16-bit dx_t of the same dtype, no float32 output, no dropout, D % 4 == 0")
        K.layernorm_bwd_s16(dy, x, w, mean, rstd, gin, dx_t, partials, dcol is not None, rows, D, dymap, xmap, dxmap)
    else:
        K.layernorm_bwd(dy, x, w, mean, rstd, gin, dx_f32, dx_t, partials, dcol is not None, rows, D, dymap, xmap, dxmap,
                        *(drop if drop else (0.0, 0)))
    assert dcol is None or dcol.dtype in (F32, w.dtype)
    if folds_deferred():        # a fused backward is collecting its folds: the three slabs join the layer's one launch (flush_folds)
        for i, out in enumerate((dw, db, dcol)):
            if out is not None:
                fold(partials[i * nblk * D:], nblk, D, D, out)
    else:
        K.layernorm_bwd_finalize(partials, nblk, D, dw, db, dcol, K.dt(w))


# ---- deferred folds ---------------------------------------------------------------------------------------------------------
# The backward of a transformer layer ends in ~4 tiny fold launches (two LayerNorm finalizes, the FeedForward bias-gradient column sums)
# whose outputs are parameter gradients nobody reads before the optimizer / the all-reduce.  Inside `with deferred_folds():` they are
# queued (the partial rows stay referenced) and go out as ONE vitk_fold_many launch per flush_folds() -- per layer in engine.TransformerFn,
# before the data-parallel sink is told the layer is done.  Same additions in the same order: bit-identical gradients.
import threading as _threading

_FOLDS = _threading.local()


def folds_deferred() -> bool:
    return getattr(_FOLDS, "q", None) is not None


def flush_folds():
    q = getattr(_FOLDS, "q", None)
    if q:
        K.fold_many(q)
        del q[:]


class deferred_folds:
    def __enter__(self):
        self.prev = getattr(_FOLDS, "q", None)
        _FOLDS.q = []
        return self

    def __exit__(self, exc_type, exc, tb):
        try:
            if exc_type is None:
                flush_folds()
        finally:
            _FOLDS.q = self.prev
        return False


def fold(part: Tensor, nparts: int, ld: int, cols: int, out: Tensor, accumulate: bool = False):
    """out[c] (+)= sum_p part[p * ld + c]: K.colsum_partials now, or a job of the enclosing deferred_folds() block."""
    q = getattr(_FOLDS, "q", None)
    if q is None:
        K.colsum_partials(part, nparts, ld, cols, out, accumulate)
    else:
        q.append((part, nparts, ld, cols, out, accumulate))


class stream_policy:
    """Per-thread override of the residual-stream dtypes for the calls made inside it (forward: fwd_stream_16, backward: grad_stream_16 --
    the fused stages read the backward's at FORWARD time and keep it on their ctx).  functional.autocast_aware uses it: the reference under
    torch.autocast keeps a float32 residual stream (LayerNorm outputs and `attn(x) + x` are float32 there, vit.py:80-81 under autocast)."""
    _tls = threading.local()

    def __init__(self, fwd16=None, grad16=None):
        self.new = (fwd16, grad16)

    def __enter__(self):
        self.old = getattr(stream_policy._tls, "v", (None, None))
        stream_policy._tls.v = self.new
        return self

    def __exit__(self, *exc):
        stream_policy._tls.v = self.old
        return False

    @staticmethod
    def current():
        return getattr(stream_policy._tls, "v", (None, None))


def grad_stream_16() -> bool:
    """The backward's residual stream (the gradient that flows through the `+ x` of vit.py:80-81) in the parameter dtype -- what
    torch autograd does when the reference runs in bfloat16 -- instead of float32: the LayerNorm backward, an HBM-bound kernel,
    moves 386 instead of 619 MB per launch (DESIGN_HISTORY.md, round 3).  On by default for 16-bit parameters without active
    dropout; VITK_GRAD_STREAM=f32 keeps the float32 stream (the forward stream is float32 either way)."""
    ov = stream_policy.current()[1]
    if ov is not None:
        return bool(ov)
    return os.environ.get("VITK_GRAD_STREAM", "16") != "f32"


def fwd_stream_16(T=None) -> bool:
    """The FORWARD residual stream x (vit.py:80-81) in the parameter dtype instead of float32 -- what the reference itself does when it
    runs in bfloat16: halves the bytes of the stream in the two residual GEMM epilogues and in the LayerNorm forward / backward reads.
    Default since round 4 for bfloat16 parameters ([measured, profiles/r04_stream16_ab.log] interleaved A/B on one box: 37.00 -> 36.20
    ms per ViT-B/16 step; logits / gradients stay inside the 1.5x-of-the-reference's-own-bf16-error gate at full depth); IEEE half
    keeps the float32 stream (its gate is an absolute one, and a half stream can overflow where the f32 stream cannot).
    VITK_FWD_STREAM=f32 forces float32, =16 forces the parameter dtype for either 16-bit type."""
    ov = stream_policy.current()[0]
    if ov is not None:
        return bool(ov)
    v = os.environ.get("VITK_FWD_STREAM", "auto")
    if v == "16":
        return True
    if v == "f32":
        return False
    return T is torch.bfloat16


def stream16_ok(M: int, D: int, I: int, Fh: int) -> bool:
    """Both residual GEMMs of a layer -- (M, D) = o (M, I) . Wout^T and (M, D) = act (M, Fh) . W2^T -- on the persistent NT kernel."""
    return I % 32 == 0 and Fh % 32 == 0 and D % 4 == 0 and _persistent_nt(M, D, I) and _persistent_nt(M, D, Fh)


# ---- column sums (bias / pos / cls gradients) -----------------------------------------------------
def colsum(x: Tensor, rows: int, cols: int, out: Tensor, accumulate: bool = False):
    ws = empty((K.colsum_ws_floats(rows, cols),), F32, x)
    K.colsum(x, rows, cols, cols, out, ws, accumulate)


# ---- Linear --------------------------------------------------------------------------------------
def _fast_nt(x: Tensor, N: int, Kd: int) -> bool:
    return x.dtype in HALF and Kd % 32 == 0 and N % 4 == 0


def fused_dropout_ok(x_dtype, M: int, N: int, Kd: int) -> bool:
    """nn.Dropout fused into a GEMM epilogue exists in the 256-row kernel only."""
    return x_dtype in HALF and Kd % 64 == 0 and K.gemm_nt_colsum_rows(M, N, Kd, N) > 0


def fp8_gemm_ok(M: int, D: int, I: int, Fh: int) -> bool:
    """The e4m3-operand GEMM (256-row kernel only; K % 64 == 0, leading dimensions % 16 == 0) serves QKV, FF1 and FF2."""
    return (D % 64 == 0 and Fh % 64 == 0 and K.gemm_nt_colsum_rows(M, 3 * I, D, 3 * I) > 0 and K.gemm_nt_colsum_rows(M, Fh, D, Fh) > 0
            and K.gemm_nt_colsum_rows(M, D, Fh, D) > 0)


def _fp8_shape_ok(M: int, N: int, Kd: int) -> bool:
    return Kd % 64 == 0 and N % 16 == 0 and K.gemm_nt_fp8_colsum_rows(M, N, Kd, N) > 0


def fp8_out_ok(M: int, D: int, I: int) -> bool:
    """The out-projection (M, D) = o (M, I) . Wout^T on e4m3 operands (256-row kernel)."""
    return _fp8_shape_ok(M, D, I)


def fp8_bwd_ok(M: int, D: int, I: int, Fh: int) -> bool:
    """The four dX GEMMs of a layer on e5m2 x e4m3 operands: (M, Fh) = g . W2, (M, D) = dpre . W1, (M, I) = g2 . Wout, (M, D) = dqkv . Wqkv."""
    return (_fp8_shape_ok(M, Fh, D) and _fp8_shape_ok(M, D, Fh) and _fp8_shape_ok(M, I, D) and _fp8_shape_ok(M, D, 3 * I))


def fp8_tn_ok(M: int, N: int, Kd: int) -> bool:
    """The fp8 weight-gradient GEMM (gemm_tn_fp8.hip) serves this (tokens, out features, in features) shape."""
    return K.gemm_tn_fp8_splits(M, N, Kd) > 0


# ---- f32 validation mode on the production MFMA kernels ------------------------------------------------------------------------
# The float32 mode (params float32: the mode that proves the host logic and the kernels' arithmetic to round-off) used to run every
# Linear on the VALU coverage kernel, i.e. it validated different kernels than the ones production runs.  For shapes the production
# kernels serve, an f32 GEMM is now ONE launch of the SAME 16-bit MFMA kernel (persistent NT kernel / split-M TN kernel) over a
# six-fold reduction extent: both operands are split into three bfloat16 terms (vitk_split_bf16x3) and the six cross products that
# matter are summed by the kernel's own f32 accumulation -- tiling, LDS swizzles, DMA ring, epilogue addressing and the split-M
# reduction are exactly the production ones; the result is f32-accurate (~2^-22).  VITK_F32_MFMA=0 restores the coverage kernel.
import os as _os

_SPLIT_CACHE = {}       # id(W) -> (weakref, weight_key, operand_b, six-block image): weights are split once per value


def f32_on_mfma() -> bool:
    return _os.environ.get("VITK_F32_MFMA", "1") not in ("0", "")


def _split_k(x: Tensor, rows: int, cols: int, ld: int, operand_b: bool) -> Tensor:
    """(rows, cols) f32 -> (rows, 6 cols) bf16: the six blocks side by side along the reduction (K) extent."""
    out = empty((rows, 6 * cols), BF16, x)
    K.split_bf16x3(x, ld, out, 6 * cols, cols, rows, cols, operand_b)
    return out


def _split_weight_k(W: Tensor) -> Tensor:
    key = weight_key(W)
    ent = _SPLIT_CACHE.get(id(W))
    if ent is not None and ent[0]() is W and ent[1] == key and not torch.cuda.is_current_stream_capturing():
        return ent[2]
    N, Kd = W.shape
    out = _split_k(W, N, Kd, Kd, True)
    if isinstance(W, torch.nn.Parameter) and not torch.cuda.is_current_stream_capturing():
        wid = id(W)
        _SPLIT_CACHE[wid] = (weakref.ref(W, lambda _r, wid=wid: _SPLIT_CACHE.pop(wid, None)), key, out)
    return out


def _f32_nt_ok(x: Tensor, M: int, N: int, Kd: int) -> bool:
    return (x.dtype == F32 and f32_on_mfma() and Kd % 32 == 0 and N % 8 == 0 and K.gemm_nt_plan(M, N, 6 * Kd, N)["persistent"])


def _gemm_nt_f32(x: Tensor, W: Tensor, M: int, resid: Optional[Tensor], bias: Optional[Tensor], w_is_split: bool = False) -> Tensor:
    """out (M, N) f32 = x @ W^T (+ resid) (+ bias) on the persistent NT kernel (f32 residual epilogue; a zero residual when there is none)."""
    N = W.shape[0]
    Kd = W.shape[1] // 6 if w_is_split else W.shape[1]
    a6 = _split_k(x, M, Kd, Kd, False)
    w6 = W if w_is_split else _split_weight_k(W)
    out = empty((M, N), F32, x)
    r = resid if resid is not None else torch.zeros((M, N), dtype=F32, device=x.device)
    K.gemm_nt_bf16(a6, 6 * Kd, w6, 6 * Kd, out, N, M, N, 6 * Kd, L.EPI_RESID, resid=r)
    if bias is not None:
        z = r if resid is None else torch.zeros((M, N), dtype=F32, device=x.device)
        K.add_rows(out, z, bias, out, M, N)
    return out


def gelu_dg_ok(T, M: int, Fh: int, D: int) -> bool:
    """The FeedForward pair FF1 (M, Fh) = a (M, D) . W1^T and dFF1 (M, Fh) = dy (M, D) . W2 can store / consume the gelu' FACTOR instead of
    the pre-activation (EPI_BIAS_GELU_DG / EPI_MUL_AUX: both on the persistent NT kernel).  Round 4: FF1's epilogue holds Phi(pre) anyway,
    one exp2 more per element gives gelu'(pre); the backward GEMM's epilogue then multiplies instead of evaluating the polynomial and the
    exponential again (it was the slowest NT kernel of the step, VALU-bound in its epilogue).  VITK_GELU_DG=0 switches it off.
    Round 5: the factor is stored as 8-bit fixed-point codes (EPI_BIAS_GELU_DG8 / EPI_MUL_AUX8, include/vitk.h: |error| <= 0.0025 over
    gelu''s whole range [-0.129, 1.129], a bf16 is off by up to 0.0039 at the top of it) -- both epilogues run at the memory system's
    rate and the factor was a third of their bytes; bfloat16 models only (float16's 11-bit factor is finer than the codes); VITK_GELU_DG=16
    keeps the 16-bit factor."""
    return (T in HALF and os.environ.get("VITK_GELU_DG", "1") != "0" and D % 32 == 0 and Fh % 32 == 0 and _persistent_nt(M, Fh, D)
            and K.gemm_nt_colsum_rows(M, Fh, D, Fh) > 0)


def gelu_dg_bits() -> int:
    """Storage of the gelu' factor between FF1 and dFF1: 8 (fixed-point codes, the default) or 16 (the model's 16-bit type)."""
    return 16 if os.environ.get("VITK_GELU_DG", "1") == "16" else 8


def linear_fwd(x: Tensor, W: Tensor, bias: Optional[Tensor], M: int, *, gelu: bool = False,
               resid: Optional[Tensor] = None, out_dtype=None, drop: Optional[Tuple[float, int]] = None, save_dg: bool = False):
    """y = x @ W^T + b  (vit.py:20,23,44,47,102).  x: (M,K) T contiguous; W: (N,K) T.

    gelu=True  -> returns (gelu(y), y)         (vit.py:20-21 fused); with save_dg (caller checked gelu_dg_ok): (gelu(y), gelu'(y))
    resid      -> returns resid + y as float32 (the `+ x` of vit.py:80-81 fused), new buffer
    """
    N, Kd = W.shape
    T = x.dtype
    if save_dg:
        if not (gelu and drop is None and bias is not None and resid is None and gelu_dg_ok(T, M, N, Kd)):
            raise L.VitkError("linear_fwd: save_dg needs gelu, a bias, no dropout and a shape gelu_dg_ok accepts")
        Wn, ldw = nt_weight(W, M, False)
        act = empty((M, N), T, x)
        if gelu_dg_bits() == 8 and T == BF16 and N % 8 == 0:      # (a float16 model keeps its 11-bit factor: the codes are at a bfloat16's accuracy)
            dg = empty((M, N), torch.uint8, x)
            K.gemm_nt_bf16(x, Kd, Wn, ldw, act, N, M, N, Kd, L.EPI_BIAS_GELU_DG8, bias=bias, aux=dg)
        else:
            dg = empty((M, N), T, x)
            K.gemm_nt_bf16(x, Kd, Wn, ldw, act, N, M, N, Kd, L.EPI_BIAS_GELU_DG, bias=bias, aux=dg)
        return act, dg
    if drop is not None and not fused_dropout_ok(T, M, N, Kd):
        raise L.VitkError("linear_fwd: fused dropout needs a shape served by the 256-row kernel (caller must check fused_dropout_ok)")
    if drop is None and _f32_nt_ok(x, M, N, Kd) and (out_dtype is None or out_dtype == F32):
        y = _gemm_nt_f32(x, W, M, resid, bias)
        if gelu:
            act = empty((M, N), F32, x)
            K.gelu_fwd(y, act)
            return act, y
        return y
    Wn, ldw = nt_weight(W, M, False) if _fast_nt(x, N, Kd) else (W, Kd)
    if resid is not None and resid.dtype in HALF:
        # 16-bit forward residual stream (fwd_stream_16): T(resid + y), the sum formed in f32 inside the epilogue
        if drop is not None or not (_fast_nt(x, N, Kd) and _persistent_nt(M, N, Kd)):
            raise L.VitkError("linear_fwd: a 16-bit residual needs a shape the persistent NT kernel serves and no fused dropout (caller must check stream16_ok)")
        out = empty((M, N), T, x)
        K.gemm_nt_bf16(x, Kd, Wn, ldw, out, N, M, N, Kd, L.EPI_RESID16, bias=bias, resid=resid)
        return out
    if resid is not None:
        out = empty((M, N), F32, x)
        if drop is not None:
            K.gemm_nt_bf16_drop(x, Kd, Wn, ldw, out, N, M, N, Kd, L.EPI_RESID, drop[0], drop[1], bias=bias, resid=resid)
        elif _fast_nt(x, N, Kd):
            K.gemm_nt_bf16(x, Kd, Wn, ldw, out, N, M, N, Kd, L.EPI_RESID, bias=bias, resid=resid)
        else:
            y = empty((M, N), T, x)
            K.gemm_generic(K.mat(x, Kd, 1), K.mat(W, 1, Kd), K.mat(y, N, 1), M, N, Kd, bias=bias)
            K.add_rows(resid, y, None, out, M, N)
        return out
    if gelu:
        act = empty((M, N), T, x)
        pre = empty((M, N), T, x)
        if drop is not None:
            if bias is None:
                raise L.VitkError("linear_fwd: the fused GELU + dropout epilogue needs a bias")
            K.gemm_nt_bf16_drop(x, Kd, Wn, ldw, act, N, M, N, Kd, L.EPI_BIAS_GELU, drop[0], drop[1], bias=bias, aux=pre)
        elif _fast_nt(x, N, Kd) and bias is not None:
            K.gemm_nt_bf16(x, Kd, Wn, ldw, act, N, M, N, Kd, L.EPI_BIAS_GELU, bias=bias, aux=pre)
        else:
            K.gemm_generic(K.mat(x, Kd, 1), K.mat(W, 1, Kd), K.mat(pre, N, 1), M, N, Kd, bias=bias)
            K.gelu_fwd(pre, act)
        return act, pre
    y = empty((M, N), out_dtype or T, x)
    if _fast_nt(x, N, Kd) and y.dtype in HALF:
        K.gemm_nt_bf16(x, Kd, Wn, ldw, y, N, M, N, Kd, L.EPI_BIAS if bias is not None else L.EPI_NONE, bias=bias)
    else:
        K.gemm_generic(K.mat(x, Kd, 1), K.mat(W, 1, Kd), K.mat(y, N, 1), M, N, Kd, bias=bias)
    return y


_WT_CACHE = {}      # id(W) -> (weakref to W, weight_key, pad, Wt); entries die with their parameter
_WP_CACHE = {}      # id(W) -> [weakref to W, weight_key, packed for y = x W^T, packed for dX = dY W]
_PERSIST_OK = {}    # (M, N, K) -> the persistent NT kernel serves the shape (vitk_gemm_nt_plan)


def _persistent_nt(M: int, N: int, Kd: int) -> bool:
    key = (M, N, Kd)
    v = _PERSIST_OK.get(key)
    if v is None:
        v = _PERSIST_OK[key] = bool(K.gemm_nt_plan(M, N, Kd, N)["persistent"])
    return v


PACK_W = True       # K-blocked weight copies for the persistent NT kernels (vitk_pack_w_nt); tests flip it to hold the row-major path against it


def packed_weights_on() -> bool:
    return PACK_W


def is_weight(W: Tensor) -> bool:
    """A tensor whose derived copies (K-blocked packs, W^T) are worth making and keeping: a Parameter, or the 16-bit copy of one that
    functional.autocast_aware made for this forward (tagged `_vitk_weight`; its cache entries die with it at the end of the step)."""
    return isinstance(W, torch.nn.Parameter) or getattr(W, "_vitk_weight", False)


def nt_weight(W: Tensor, M: int, transposed: bool):
    """(operand, ldw) for the NT GEMM whose "W" is this weight: `transposed=False` -- y = x W^T (rows N, reduction K);
    `transposed=True` -- dX = dY W (rows K, reduction N).  For shapes the persistent kernel serves, a K-blocked copy
    (vitk_pack_w_nt, ldw = 0) cached per parameter VALUE like the transposed copies (weight_key): the kernel's LDS-DMA then
    reads whole 128-byte lines of W (the 8 GEMM shapes of a ViT-B/16 layer: 1.70 -> 1.59 ms).  Both copies are made by one
    call on the first forward use while the CALLER has gradients enabled (_epoch.caller_grad_mode); never cached during HIP-graph
    capture.  Raw `.data` writes to a parameter need `vit_pytorch_amd.invalidate_weight_caches()` (they do not bump `_version`)."""
    N, Kd = W.shape
    rows, red = (Kd, N) if transposed else (N, Kd)
    if (W.dtype in HALF and is_weight(W) and W.is_contiguous() and red % 32 == 0 and packed_weights_on()
            and _persistent_nt(M, rows, red)):
        capturing = torch.cuda.is_current_stream_capturing()
        key = weight_key(W)
        ent = _WP_CACHE.get(id(W))
        if capturing or ent is None or ent[0]() is not W or ent[1] != key:
            wid = id(W)
            ent = [weakref.ref(W, lambda _r, wid=wid: _WP_CACHE.pop(wid, None)), key, None, None]
            if not capturing:
                _WP_CACHE[wid] = ent
        idx = 3 if transposed else 2
        if ent[idx] is None:
            want_f = not transposed and Kd % 32 == 0
            # (the caller's grad mode: torch.is_grad_enabled() is always False inside autograd.Function.forward, where this runs)
            want_t = (transposed or (caller_grad_mode() and W.requires_grad and ent[3] is None)) and N % 32 == 0
            pf = empty((K.pack_w_nt_bytes(N, Kd) // 2,), W.dtype, W) if want_f else None
            pt = empty((K.pack_w_nt_bytes(Kd, N) // 2,), W.dtype, W) if want_t else None
            K.pack_w_nt(W, Kd, N, Kd, pf, pt)
            if pf is not None:
                ent[2] = pf
            if pt is not None:
                ent[3] = pt
        return ent[idx], 0
    if transposed:
        return transpose_weight(W), N
    return W, Kd


def prepack_weights(Ws, M: int) -> int:
    """The K-blocked copies nt_weight() would make one by one during this step -- forward pack and, when the caller wants gradients,
    the transposed pack of every weight in `Ws` whose cache entry is missing or stale -- in ONE table-driven launch per 96 packs
    (vitk_pack_w_nt_many): a training step re-packs every Linear of the stack after each optimizer step, 8 launches a layer before.
    Same conditions, same cache entries as nt_weight(); returns the number of packs made."""
    if not packed_weights_on() or torch.cuda.is_current_stream_capturing():
        return 0
    rows, ents = [], []
    want_grad = caller_grad_mode()
    for W in Ws:
        if W is None or W.dtype not in HALF or not is_weight(W) or not W.is_contiguous():
            continue
        N, Kd = W.shape
        key = weight_key(W)
        ent = _WP_CACHE.get(id(W))
        if ent is None or ent[0]() is not W or ent[1] != key:
            wid = id(W)
            ent = [weakref.ref(W, lambda _r, wid=wid: _WP_CACHE.pop(wid, None)), key, None, None]
            _WP_CACHE[wid] = ent
        want_f = ent[2] is None and Kd % 32 == 0 and _persistent_nt(M, N, Kd)
        want_t = ent[3] is None and want_grad and W.requires_grad and N % 32 == 0 and _persistent_nt(M, Kd, N)
        if not (want_f or want_t):
            continue
        pf = empty((K.pack_w_nt_bytes(N, Kd) // 2,), W.dtype, W) if want_f else None
        pt = empty((K.pack_w_nt_bytes(Kd, N) // 2,), W.dtype, W) if want_t else None
        rows.append((W, N, Kd, pf, pt)); ents.append(ent)
    if not rows:
        return 0
    K.pack_w_nt_many(rows)
    for (W, N, Kd, pf, pt), ent in zip(rows, ents):
        if pf is not None:
            ent[2] = pf
        if pt is not None:
            ent[3] = pt
    return sum((r[3] is not None) + (r[4] is not None) for r in rows)


def transpose_weight(W: Tensor, pad_to: int = 0) -> Tensor:
    """W (N, K) -> W^T (K, N) (optionally with zero columns up to pad_to): the operand that makes dX = dY . W an NT GEMM.
    Cached per parameter until its values change (weight_key: data_ptr, torch version counter, and the epoch this package's
    own optimizer bumps) -- a training step used to re-transpose every weight (0.28 ms of the ViT-B/16 step).  Never cached
    while a HIP graph is being captured: the transpose must then be part of the graph so that replays see current weights."""
    N, Kd = W.shape
    capturing = torch.cuda.is_current_stream_capturing()
    key = weight_key(W)
    ent = _WT_CACHE.get(id(W))
    if not capturing and ent is not None and ent[0]() is W and ent[1] == key and ent[2] == pad_to:
        return ent[3]
    Wt = empty((Kd, N), W.dtype, W)
    K.transpose(W, Wt, N, Kd)
    if pad_to and pad_to > N:
        Wt = pad_cols(Wt, Kd, N, pad_to)
    if not capturing and is_weight(W):
        wid = id(W)
        _WT_CACHE[wid] = (weakref.ref(W, lambda _r, wid=wid: _WT_CACHE.pop(wid, None)), key, pad_to, Wt)
    return Wt


def linear_dx(dy: Tensor, W: Tensor, M: int, *, gelu_pre: Optional[Tensor] = None, db: Optional[Tensor] = None,
              drop: Optional[Tuple[float, int]] = None, gelu_dg: Optional[Tensor] = None):
    """dX = dY @ W  (optionally * gelu'(pre): the GELU backward fused as an epilogue; gelu_dg = that factor itself, stored by a
    linear_fwd(save_dg=True) forward: the epilogue only multiplies).

    With gelu_pre / gelu_dg and db: returns (dX, done) -- done is True when colsum(dX), the bias gradient of the Linear that
    produced `pre`, was written to db as a by-product of the GEMM epilogue (otherwise the caller still owes it)."""
    N, Kd = W.shape
    T = dy.dtype
    if gelu_dg is not None:
        if gelu_pre is not None or drop is not None or not gelu_dg_ok(T, M, Kd, N):
            raise L.VitkError("linear_dx: gelu_dg goes without gelu_pre / dropout and with a shape gelu_dg_ok accepts")
        dx = empty((M, Kd), T, dy)
        Wt, ldt = nt_weight(W, M, True)
        R = K.gemm_nt_colsum_rows(M, Kd, N, Kd, T)
        part = empty((R * Kd,), F32, dy) if db is not None else None
        if gelu_dg.dtype == torch.uint8:        # the codes an EPI_BIAS_GELU_DG8 forward stored
            K.gemm_nt_bf16_mul_aux8_colsum(dy, N, Wt, ldt, dx, Kd, M, Kd, N, gelu_dg, part)
        else:
            K.gemm_nt_bf16_mul_aux_colsum(dy, N, Wt, ldt, dx, Kd, M, Kd, N, gelu_dg, part)
        if db is not None:
            fold(part, R, Kd, Kd, db)
            return dx, True
        return dx
    if drop is None and _f32_nt_ok(dy, M, Kd, N):
        dx = _gemm_nt_f32(dy, transpose_weight(W), M, None, None)          # (M, N) @ (Kd, N)^T
        if gelu_pre is not None:
            K.gelu_bwd(dx, gelu_pre, dx)
        return (dx, False) if db is not None else dx
    dx = empty((M, Kd), T, dy)
    if T in HALF and N % 32 == 0 and Kd % 4 == 0:
        Wt, ldt = nt_weight(W, M, True)  # W^T (K, N): makes dX an NT GEMM with reduction dim N contiguous (K-blocked copy or plain transpose)
        if drop is not None:        # backward of dropout(gelu(pre)): same keep decisions, fused with GELU' (and db)
            R = K.gemm_nt_colsum_rows(M, Kd, N, Kd, T)
            if gelu_pre is None or R == 0:
                raise L.VitkError("linear_dx: fused dropout backward needs gelu_pre and a shape served by the 256-row kernel")
            part = empty((R * Kd,), F32, dy) if db is not None else None
            K.gemm_nt_bf16_drop(dy, N, Wt, ldt, dx, Kd, M, Kd, N, L.EPI_GELU_BWD, drop[0], drop[1], aux=gelu_pre, partials=part)
            if db is not None:
                fold(part, R, Kd, Kd, db)
                return dx, True
            return dx
        if gelu_pre is not None and db is not None:
            R = K.gemm_nt_colsum_rows(M, Kd, N, Kd, T)
            if R > 0:
                part = empty((R * Kd,), F32, dy)
                K.gemm_nt_bf16_gelu_bwd_colsum(dy, N, Wt, ldt, dx, Kd, M, Kd, N, gelu_pre, part)
                fold(part, R, Kd, Kd, db)
                return dx, True
        if gelu_pre is not None:
            K.gemm_nt_bf16(dy, N, Wt, ldt, dx, Kd, M, Kd, N, L.EPI_GELU_BWD, aux=gelu_pre)
        else:
            K.gemm_nt_bf16(dy, N, Wt, ldt, dx, Kd, M, Kd, N)
    else:
        K.gemm_generic(K.mat(dy, N, 1), K.mat(W, Kd, 1), K.mat(dx, Kd, 1), M, Kd, N)
        if gelu_pre is not None:
            K.gelu_bwd(dx, gelu_pre, dx)
    return (dx, False) if db is not None else dx


def linear_dw(dy: Tensor, x: Tensor, M: int, dW: Tensor, db: Optional[Tensor] = None,
              ldy: Optional[int] = None, ldx: Optional[int] = None):
    """dW = dY^T X (reduced over the M token rows), db = colsum(dY).  dW: (N,K) of dtype T."""
    N, Kd = dW.shape
    ldy = ldy or N
    ldx = ldx or Kd
    if (dy.dtype == F32 and f32_on_mfma() and dW.dtype == F32 and N % 8 == 0 and Kd % 8 == 0 and ldy % 4 == 0 and ldx % 4 == 0
            and M >= 4096 and N >= 256 and Kd >= 256):
        # f32 validation mode on the production split-M TN kernel: the six bf16 blocks stacked along the reduction (token) extent
        y6 = empty((6 * M, N), BF16, dy); x6 = empty((6 * M, Kd), BF16, dy)
        K.split_bf16x3(dy, ldy, y6, N, M * N, M, N, False)
        K.split_bf16x3(x, ldx, x6, Kd, M * Kd, M, Kd, True)
        splits = K.gemm_tn_splits(6 * M, N, Kd)
        ws = empty((splits * N * Kd,), F32, dy)
        K.gemm_tn_bf16(y6, N, x6, Kd, dW, Kd, 6 * M, N, Kd, ws, splits)
    elif dy.dtype in HALF and N % 8 == 0 and Kd % 8 == 0 and ldy % 8 == 0 and ldx % 8 == 0:
        splits = K.gemm_tn_splits(M, N, Kd, dy.dtype)
        ws = empty((splits * N * Kd,), F32, dy)
        K.gemm_tn_bf16(dy, ldy, x, ldx, dW, Kd, M, N, Kd, ws, splits)
    else:
        K.gemm_generic(K.mat(dy, 1, ldy), K.mat(x, ldx, 1), K.mat(dW, Kd, 1), N, Kd, M)
    if db is not None:
        assert ldy == N
        colsum(dy, M, N, db)


def linear_dw_pair(dy0: Tensor, x0: Tensor, dW0: Tensor, dy1: Tensor, x1: Tensor, dW1: Tensor, M: int):
    """Two weight gradients over the same token rows (to_out's and to_qkv's of a layer) -- ONE launch of the split-M kernel where the
    pair is served (K.gemm_tn_pair_splits: both large, 16-bit; the library switch that turns pairing off is listed in README.md): 36 tiles x 7 splits at ViT-B/16 instead of 27 x 9
    and 9 x 28, i.e. half the f32 slabs, one launch and one fold less; otherwise two linear_dw calls."""
    (N0, K0), (N1, K1) = dW0.shape, dW1.shape
    if dy0.dtype in HALF and dy1.dtype == dy0.dtype and dW0.dtype == dW1.dtype and dW0.dtype in HALF + (F32,):
        splits = K.gemm_tn_pair_splits(M, N0, K0, N1, K1, dy0.dtype)
        if splits > 0:
            ws = empty((splits * (N0 * K0 + N1 * K1),), F32, dy0)
            K.gemm_tn_bf16_pair(dy0, N0, x0, K0, dW0, dy1, N1, x1, K1, dW1, M, ws, splits)
            return
    linear_dw(dy0, x0, M, dW0)
    linear_dw(dy1, x1, M, dW1)


def pad_cols(x: Tensor, rows: int, cols: int, cols_pad: int) -> Tensor:
    """(rows, cols) -> (rows, cols_pad) with zero columns appended (K padding for the MFMA GEMMs)."""
    out = empty((rows, cols_pad), x.dtype, x)
    K.copy_cols(x, cols, out, cols_pad, rows, cols, cols_pad)
    return out


def unpad_cols(x: Tensor, rows: int, cols_pad: int, cols: int, out: Optional[Tensor] = None) -> Tensor:
    out = out if out is not None else empty((rows, cols), x.dtype, x)
    K.copy_cols(x, cols_pad, out, cols, rows, cols, cols)
    return out


# ---- attention core --------------------------------------------------------------------------------
def attn_fast_ok(T, N: int, d: int) -> bool:
    """whole-head-in-LDS kernels: bf16, dim_head 64, N <= 480"""
    return T in HALF and d == 64 and 1 <= N <= 480


def attn_x2_ok(T, N: int, d: int) -> bool:
    """f32 validation mode on the FLASH kernels (round 3): operands split into hi + lo 16-bit terms, three MFMAs per product, f32
    outputs -- the same staging / masking / online-softmax code as the 16-bit kernels (csrc/attention_pipe.hip, NS = 2), so the
    1e-3 logic gate of the f32 mode covers them.  VITK_F32_MFMA=0 (the f32 mode on the coverage kernels) restores the materialising kernels."""
    return T == torch.float32 and d == 64 and 32 < N <= 224 and f32_on_mfma()


def _split2(x: Tensor):
    hi = empty(tuple(x.shape), torch.bfloat16, x); lo = empty(tuple(x.shape), torch.bfloat16, x)
    K.split2(x, hi, lo)
    return hi, lo


def attn_varlen_ok(T, d: int) -> bool:
    """chunked flash kernels (any N): 16-bit, dim_head 32 / 48 / 64 / 80 / 96 (ViT-H/14: dim_head 80, N = 577; vit.py:86 leaves dim_head
    free -- the other widths a multiple of 16 up to 96 were instantiated in round 3)."""
    return T in HALF and d in (32, 48, 64, 80, 96)


def attn_fwd(qkv: Tensor, B: int, N: int, H: int, d: int, scale: float, drop: Optional[Tuple[float, int]] = None):
    """softmax(scale * q k^T) v on the merged (B*N, 3*H*d) to_qkv output (vit.py:54-63).
    Returns (o (B*N, H*d), saved) where saved is lse (fused) or the attention matrix P."""
    I = H * d
    T = qkv.dtype
    o = empty((B * N, I), T, qkv)
    sb, sh, sn = N * 3 * I, d, 3 * I
    if drop is not None and not attn_fast_ok(T, N, d):
        raise L.VitkError("attn_fwd: attention-matrix dropout is fused in the fixed-length kernel only (caller must check attn_fast_ok)")
    if attn_fast_ok(T, N, d):
        lse = empty((B, H, N), F32, qkv)
        K.attn_fwd_bf16(K.bhnd(qkv, sb, sh, sn), K.bhnd(qkv, sb, sh, sn, offset=I), K.bhnd(qkv, sb, sh, sn, offset=2 * I),
                        K.bhnd(o, N * I, d, I), lse, B, H, N, d, scale, *(drop if drop else (0.0, 0)))
        return o, lse
    if attn_x2_ok(T, N, d):
        hi, lo = _split2(qkv)
        pair = lambda off: (K.bhnd(hi, sb, sh, sn, offset=off), K.bhnd(lo, sb, sh, sn, offset=off))
        lse = empty((B, H, N), F32, qkv)
        K.attn_fwd_x2(pair(0), pair(I), pair(2 * I), K.bhnd(o, N * I, d, I), lse, B, H, N, d, scale)
        return o, lse
    if attn_varlen_ok(T, d):
        sg = uniform_segments(B, N, qkv.device)
        lse = empty((H, B * N), F32, qkv)
        K.attn_varlen_fwd_bf16(K.hnd(qkv, d, sn), K.hnd(qkv, d, sn, offset=I), K.hnd(qkv, d, sn, offset=2 * I), K.hnd(o, d, I), lse,
                               sg.cu_q, sg.cu_k, sg.qblk_seg, sg.qblk_r0, sg.nqblk, B * N, H, d, scale)
        return o, lse
    # materialising path: S = q k^T ; P = softmax(scale*S) ; O = P v   (batched over (B, H) in place)
    S = empty((B, H, N, N), T, qkv)
    K.gemm_generic(K.mat(qkv, sn, 1, sb, sh), K.mat(qkv, 1, sn, sb, sh, offset=I), K.mat(S, N, 1, H * N * N, N * N),
                   N, N, d, nb1=B, nb2=H)
    P = empty((B, H, N, N), T, qkv)
    K.softmax_fwd(S, P, B * H * N, N, scale)
    K.gemm_generic(K.mat(P, N, 1, H * N * N, N * N), K.mat(qkv, sn, 1, sb, sh, offset=2 * I), K.mat(o, I, 1, N * I, d),
                   N, d, N, nb1=B, nb2=H)
    return o, P


def attn_bwd(qkv: Tensor, o: Tensor, do: Tensor, saved: Tensor, B: int, N: int, H: int, d: int, scale: float,
             drop: Optional[Tuple[float, int]] = None) -> Tensor:
    """Returns dqkv (B*N, 3*H*d) in the merged layout (it is the dY of the to_qkv GEMM)."""
    I = H * d
    T = qkv.dtype
    dqkv = empty((B * N, 3 * I), T, qkv)
    sb, sh, sn = N * 3 * I, d, 3 * I
    if attn_fast_ok(T, N, d):
        delta = empty((B, H, N), F32, qkv)
        K.attn_bwd_bf16(K.bhnd(qkv, sb, sh, sn), K.bhnd(qkv, sb, sh, sn, offset=I), K.bhnd(qkv, sb, sh, sn, offset=2 * I),
                        K.bhnd(o, N * I, d, I), K.bhnd(do, N * I, d, I), saved, delta,
                        K.bhnd(dqkv, sb, sh, sn), K.bhnd(dqkv, sb, sh, sn, offset=I), K.bhnd(dqkv, sb, sh, sn, offset=2 * I),
                        B, H, N, d, scale, *(drop if drop else (0.0, 0)))
        return dqkv
    if attn_x2_ok(T, N, d):
        hi, lo = _split2(qkv)
        dhi, dlo = _split2(do.contiguous())
        pair = lambda off: (K.bhnd(hi, sb, sh, sn, offset=off), K.bhnd(lo, sb, sh, sn, offset=off))
        delta = empty((B, H, N), F32, qkv)
        K.attn_bwd_x2(pair(0), pair(I), pair(2 * I), K.bhnd(o, N * I, d, I), (K.bhnd(dhi, N * I, d, I), K.bhnd(dlo, N * I, d, I)), saved, delta,
                      K.bhnd(dqkv, sb, sh, sn), K.bhnd(dqkv, sb, sh, sn, offset=I), K.bhnd(dqkv, sb, sh, sn, offset=2 * I), B, H, N, d, scale)
        return dqkv
    if attn_varlen_ok(T, d):
        sg = uniform_segments(B, N, qkv.device)
        delta = empty((H, B * N), F32, qkv)
        K.attn_varlen_bwd_bf16(K.hnd(qkv, d, sn), K.hnd(qkv, d, sn, offset=I), K.hnd(qkv, d, sn, offset=2 * I), K.hnd(o, d, I),
                               K.hnd(do, d, I), saved, delta, K.hnd(dqkv, d, sn), K.hnd(dqkv, d, sn, offset=I),
                               K.hnd(dqkv, d, sn, offset=2 * I), sg.cu_q, sg.cu_k, sg.qblk_seg, sg.qblk_r0, sg.nqblk,
                               sg.kblk_seg, sg.kblk_r0, sg.nkblk, B * N, H, d, scale)
        return dqkv
    P = saved
    pm = K.mat(P, N, 1, H * N * N, N * N)
    pmT = K.mat(P, 1, N, H * N * N, N * N)
    dom = K.mat(do, I, 1, N * I, d)
    # dV = P^T dO
    K.gemm_generic(pmT, dom, K.mat(dqkv, sn, 1, sb, sh, offset=2 * I), N, d, N, nb1=B, nb2=H)
    # dP = dO V^T ; dS = scale * P * (dP - rowsum(dP*P))
    dP = empty((B, H, N, N), T, qkv)
    K.gemm_generic(dom, K.mat(qkv, 1, sn, sb, sh, offset=2 * I), K.mat(dP, N, 1, H * N * N, N * N), N, N, d, nb1=B, nb2=H)
    dS = dP
    K.softmax_bwd(P, dP, dS, B * H * N, N, scale)
    # dQ = dS K ; dK = dS^T Q
    K.gemm_generic(K.mat(dS, N, 1, H * N * N, N * N), K.mat(qkv, sn, 1, sb, sh, offset=I), K.mat(dqkv, sn, 1, sb, sh),
                   N, d, N, nb1=B, nb2=H)
    K.gemm_generic(K.mat(dS, 1, N, H * N * N, N * N), K.mat(qkv, sn, 1, sb, sh), K.mat(dqkv, sn, 1, sb, sh, offset=I),
                   N, d, N, nb1=B, nb2=H)
    return dqkv
