"""TEST INFRASTRUCTURE ONLY -- CPU doubles of the libvitk entry points the three fused stages (patch embedding, Transformer, head) call.

`installed()` swaps a few dozen functions of `vit_pytorch_amd.kernels` / `vit_pytorch_amd.ops` for torch-CPU stand-ins built from the
oracle's per-op restatements (oracle/vit_oracle.py), so that the HOST LOGIC of the drop-in -- which tensor goes into which launch
with which strides and row maps, the saved-activation bookkeeping, the fp8 slot / scale / recording state machine, the weight
caches, the data-parallel gradient sink -- can be run and checked in the CPU suite, where no kernel can launch: the fused stage
alone (tests/test_engine_host_logic.py), whole models against the reference's goldens (ViT, SimpleViT, the three sibling variants,
NaViT: tests/test_models_host_logic.py), two gloo ranks (tests/test_parallel_gloo.py).  Nothing here is importable from the product package, the product has
no CPU path (`kernels.require_device` raises on CPU tensors), and no parity claim rests on these doubles: the kernels themselves
are tested against float64 / the oracle / the reference's goldens in the `-m gpu` tests.

Each double states the contract of the entry point it stands in for (include/vitk.h) in torch: compute in float32, round to the
output dtype.
"""
from __future__ import annotations

import contextlib

import torch

from oracle import vit_oracle as O
from vit_pytorch_amd import _lib as L
from vit_pytorch_amd import kernels as K
from vit_pytorch_amd import ops

F32 = torch.float32
E4M3, E5M2 = torch.float8_e4m3fn, torch.float8_e5m2
CALLS = []          # (name, info) log of the launches a test wants to assert on


def _rec_amax(amax64, v):
    """atomicMax of the float bit pattern into word 0 of the slot's 64."""
    cur = amax64.view(torch.int32)[0:1].view(F32)
    amax64.view(torch.int32)[0:1] = torch.maximum(cur, v.abs().max().to(F32).reshape(1)).view(torch.int32)


def _q8(v, scale, fmt):
    fmax = 448.0 if fmt == E4M3 else 57344.0
    return (v.float() * scale).clamp(-fmax, fmax).to(fmt).view(torch.uint8)


def _deq(t8, fmt):
    return t8.view(fmt).float()


def _w_plain(W, ldw, N, Kd):
    """The W operand of an NT GEMM as (N, Kd) float32: row-major (ldw == Kd) or a K-blocked copy (ldw == 0: the double of
    vitk_pack_w_nt keeps the plain rows at the start of the buffer)."""
    if ldw == 0:
        return W.flatten()[:N * Kd].view(N, Kd).float()
    assert ldw == Kd
    return W.reshape(N, Kd).float()


def _epilogue(acc, C, M, N, epilogue, bias, resid, aux, partials=None):
    if epilogue in (L.EPI_BIAS, L.EPI_BIAS_GELU, L.EPI_BIAS_GELU_DG, L.EPI_BIAS_GELU_DG8, L.EPI_RESID, L.EPI_RESID16) and bias is not None:
        acc = acc + bias.float()
    if epilogue == L.EPI_BIAS_GELU_DG8:                             # aux = 8-bit codes of gelu'(pre): rne(200 f) + 27 (include/vitk.h)
        assert aux.dtype == torch.uint8
        pre = acc.to(C.dtype).float()
        aux.view(M, N).copy_((torch.round(O.gelu_bwd(torch.ones_like(pre), pre) * 200.0) + 27.0).clamp_(0, 255).to(torch.uint8))
        C.view(M, N).copy_(O.gelu_fwd(pre))
        return
    if epilogue == L.EPI_MUL_AUX8:
        assert aux.dtype == torch.uint8
        C.view(M, N).copy_(acc * ((aux.view(M, N).float() - 27.0) * 0.005))
        if partials is not None:
            partials.zero_()
            partials[:N] = C.view(M, N).float().sum(0)
        return
    if epilogue == L.EPI_BIAS_GELU_DG:                              # aux = gelu'(pre) of the pre-activation rounded to T
        pre = acc.to(aux.dtype).float()
        aux.view(M, N).copy_(O.gelu_bwd(torch.ones_like(pre), pre))
        C.view(M, N).copy_(O.gelu_fwd(pre))
    elif epilogue == L.EPI_MUL_AUX:
        C.view(M, N).copy_(acc * aux.view(M, N).float())
        if partials is not None:
            partials.zero_()
            partials[:N] = C.view(M, N).float().sum(0)
    elif epilogue == L.EPI_BIAS_GELU:
        aux.view(M, N).copy_(acc)                                   # pre-activation, rounded to T
        if C is not None:                                           # (fp8, lean saving: only the e4m3 copy is wanted)
            C.view(M, N).copy_(O.gelu_fwd(aux.view(M, N).float()))
    elif epilogue == L.EPI_RESID:
        assert C.dtype == F32 and resid.dtype == F32
        C.view(M, N).copy_(acc + resid.view(M, N))
    elif epilogue == L.EPI_RESID16:
        assert C.dtype == resid.dtype and C.dtype != F32
        C.view(M, N).copy_(acc + resid.view(M, N).float())
    elif epilogue == L.EPI_GELU_BWD:
        C.view(M, N).copy_(O.gelu_bwd(acc, aux.view(M, N).float()))
        if partials is not None:
            partials.zero_()
            partials[:N] = C.view(M, N).float().sum(0)              # of the ROUNDED values, like the kernel
    else:
        C.view(M, N).copy_(acc)


def _rows(t, rows, cols, ld):
    """(rows, cols) view of a tensor whose logical rows are `ld` elements apart (row 0 of every image of a (B, N, D) tensor: ld = N D)."""
    return torch.as_strided(t, (rows, cols), (ld, 1), t.storage_offset())


def gemm_nt_bf16(A, lda, W, ldw, C, ldc, M, N, Kd, epilogue=L.EPI_NONE, bias=None, resid=None, aux=None):
    CALLS.append(("gemm_nt_bf16", (M, N, Kd, epilogue)))
    acc = _rows(A, M, Kd, lda).float() @ _w_plain(W, ldw, N, Kd).t()
    if ldc == N:
        _epilogue(acc, C, M, N, epilogue, bias, resid, aux)
    else:               # strided output rows (the head's dX written into row 0 of every image): plain / bias epilogues only
        assert epilogue in (L.EPI_NONE, L.EPI_BIAS)
        _rows(C, M, N, ldc).copy_(acc + (bias.float() if (epilogue == L.EPI_BIAS and bias is not None) else 0))


def gemm_nt_bf16_gelu_bwd_colsum(A, lda, W, ldw, C, ldc, M, N, Kd, aux, partials):
    CALLS.append(("gemm_nt_bf16_gelu_bwd_colsum", (M, N, Kd)))
    _epilogue(A.reshape(M, Kd).float() @ _w_plain(W, ldw, N, Kd).t(), C, M, N, L.EPI_GELU_BWD, None, None, aux, partials)


def gemm_nt_bf16_mul_aux_colsum(A, lda, W, ldw, C, ldc, M, N, Kd, aux, partials):
    CALLS.append(("gemm_nt_bf16_mul_aux_colsum", (M, N, Kd)))
    _epilogue(A.reshape(M, Kd).float() @ _w_plain(W, ldw, N, Kd).t(), C, M, N, L.EPI_MUL_AUX, None, None, aux, partials)


def gemm_nt_fp8_v2(A, lda, W, ldw, C, ldc, M, N, Kd, epilogue, *, a_kind, bias=None, resid=None, aux=None, partials=None, alpha=1.0,
                   alpha_a=None, alpha_w=None, c8=None, c8_scale=None, c8_amax64=None, k128=False):
    assert lda == Kd and ldw == Kd and ldc == N and Kd % 64 == 0 and (not k128 or Kd % 128 == 0)
    assert W.dtype == torch.uint8 and W.shape == (N, Kd)
    assert C is not None or (epilogue == L.EPI_BIAS_GELU and c8 is not None and a_kind != K.A_16BIT)
    CALLS.append(("gemm_nt_fp8_v2", (M, N, Kd, epilogue, a_kind, bool(k128))))
    if a_kind == K.A_16BIT:
        raise AssertionError("the doubles expect the recording pass through gemm_nt_fp8_ex")
    a = _deq(A.reshape(M, Kd), E4M3 if a_kind == K.A_E4M3 else E5M2)
    al = alpha * (float(alpha_a[0]) if alpha_a is not None else 1.0) * (float(alpha_w[0]) if alpha_w is not None else 1.0)
    acc = (a @ _deq(W, E4M3).t()) * al
    _epilogue(acc, C, M, N, epilogue, bias, resid, aux, partials)
    if epilogue == L.EPI_BIAS_GELU:
        g = O.gelu_fwd(aux.view(M, N).float()).to(aux.dtype) if C is None else C.view(M, N)
        if c8_amax64 is not None:
            _rec_amax(c8_amax64, g.float())
        if c8 is not None:
            c8.view(M, N).copy_(_q8(g, float(c8_scale[0]), E4M3))
    elif epilogue == L.EPI_GELU_BWD:                                # e5m2 copy of the gradient the epilogue just formed
        if c8_amax64 is not None:
            _rec_amax(c8_amax64, C.view(M, N).float())
        if c8 is not None:
            c8.view(M, N).copy_(_q8(C.view(M, N), float(c8_scale[0]), E5M2))
    else:
        assert c8 is None and c8_amax64 is None


def gemm_nt_fp8_ex(A, lda, W, ldw, C, ldc, M, N, Kd, epilogue, *, a_is_fp8, bias=None, resid=None, aux=None, alpha=1.0, alpha_a=None,
                   alpha_w=None, c8=None, c8_scale=None, c8_amax64=None):
    if a_is_fp8:
        return gemm_nt_fp8_v2(A, lda, W, ldw, C, ldc, M, N, Kd, epilogue, a_kind=K.A_E4M3, bias=bias, resid=resid, aux=aux, alpha=alpha,
                              alpha_a=alpha_a, alpha_w=alpha_w, c8=c8, c8_scale=c8_scale, c8_amax64=c8_amax64)
    CALLS.append(("gemm_nt_fp8_ex/16bit", (M, N, Kd, epilogue)))
    _epilogue(A.reshape(M, Kd).float() @ W.reshape(N, Kd).float().t(), C, M, N, epilogue, bias, resid, aux)
    if epilogue == L.EPI_BIAS_GELU and c8_amax64 is not None:
        _rec_amax(c8_amax64, C.view(M, N).float())


def pack_w_nt(W, ldw, N, Kd, out, out_t):
    if out is not None:
        out[:N * Kd] = W.reshape(N, Kd).flatten()
    if out_t is not None:
        out_t[:N * Kd] = W.reshape(N, Kd).t().contiguous().flatten()


def pack_w_nt_many(rows):
    CALLS.append(("pack_w_nt_many", len(rows)))
    for W, N, Kd, out, out_t in rows:
        pack_w_nt(W, Kd, N, Kd, out, out_t)


def gemm_tn_bf16_pair(dY0, ldy0, X0, ldx0, dW0, dY1, ldy1, X1, ldx1, dW1, M, ws, splits, accumulate0=False, accumulate1=False):
    (N0, K0), (N1, K1) = dW0.shape, dW1.shape
    gemm_tn_bf16(dY0, ldy0, X0, ldx0, dW0, K0, M, N0, K0, ws, splits, accumulate0)
    gemm_tn_bf16(dY1, ldy1, X1, ldx1, dW1, K1, M, N1, K1, ws, splits, accumulate1)


def gemm_tn_bf16(dY, ldy, X, ldx, dW, ldo, M, N, Kd, ws, splits, accumulate=False):
    assert ldo == Kd
    CALLS.append(("gemm_tn_bf16", (M, N, Kd)))
    r = _rows(dY, M, N, ldy).float().t() @ _rows(X, M, Kd, ldx).float()
    dW.view(N, Kd).copy_(r + (dW.view(N, Kd).float() if accumulate else 0))


def gemm_tn_fp8(dY8, ldy, X8, ldx, dW, ldo, M, N, Kd, ws, splits, *, alpha_y=None, alpha_x=None, k128=False, accumulate=False):
    assert ldy == N and ldx == Kd and ldo == Kd and dY8.dtype == torch.uint8 and X8.dtype == torch.uint8 and splits >= 1
    assert ws.numel() >= splits * N * Kd
    CALLS.append(("gemm_tn_fp8", (M, N, Kd, bool(k128))))
    al = (float(alpha_y[0]) if alpha_y is not None else 1.0) * (float(alpha_x[0]) if alpha_x is not None else 1.0)
    r = (_deq(dY8.reshape(M, N), E5M2).t() @ _deq(X8.reshape(M, Kd), E4M3)) * al
    dW.view(N, Kd).copy_(r + (dW.view(N, Kd).float() if accumulate else 0))


def _map_rows(m, rows):
    """vitk_rowmap (include/vitk.h): logical row r -> physical row (r / group) * gstride + r % group + offset; group <= 0: identity."""
    r = torch.arange(rows)
    if m.group <= 0:
        return r
    return (r // m.group) * m.gstride + (r % m.group) + m.offset


def layernorm_fwd(x, w, b, y, mean, rstd, rows, D, eps=1e-5, imap=L.IDENT, omap=L.IDENT, add=None, add_group=0, add_off=0, y8=None,
                  scale8=None, amax64=None):
    xin = x.reshape(-1, D)[_map_rows(imap, rows)].float()
    yy, m, r = O.layer_norm_fwd(xin, w.float(), None if b is None else b.float(), eps)
    if add is not None:         # positional table added on the way out (vit.py:126): row r takes add[(r % add_group) + add_off]
        rr = torch.arange(rows)
        yy = yy + add.reshape(-1, D)[((rr % add_group) if add_group > 0 else rr) + add_off].float()
    oi = _map_rows(omap, rows)
    yv = y.view(-1, D)
    yv[oi] = yy.to(y.dtype)
    mean.copy_(m); rstd.copy_(r)
    if amax64 is not None:
        _rec_amax(amax64, yy)
    if y8 is not None:
        y8.view(-1, D)[oi] = _q8(yy, float(scale8[0]), E4M3)


def ln_bwd(dy, x, w, mean, rstd, rows, D, *, gin=None, dx_f32=None, dx_t=None, dw=None, db=None, dcol=None, dymap=L.IDENT,
           xmap=L.IDENT, dxmap=L.IDENT, drop=None):
    assert drop is None
    dyr = dy.reshape(-1, D)[_map_rows(dymap, rows)].float()
    xr = x.reshape(-1, D)[_map_rows(xmap, rows)].float()
    dx, gw, gb = O.layer_norm_bwd(dyr, xr, w.float(), mean, rstd)
    if gin is not None:
        dx = dx + gin.reshape(rows, D).float()
    oi = _map_rows(dxmap, rows)
    if dx_f32 is not None:
        dx_f32.view(-1, D)[oi] = dx
    if dx_t is not None:
        dx_t.view(-1, D)[oi] = dx.to(dx_t.dtype)
    if dw is not None:
        dw.copy_(gw)
    if db is not None:
        db.copy_(gb)
    if dcol is not None:
        dcol.copy_((dx_t.view(-1, D)[oi].float() if dx_t is not None else dx).sum(0))


def _heads(qkv, B, N, H, d):
    I = H * d
    t = qkv.reshape(B, N, 3, H, d).float()
    return (t[:, :, i].permute(0, 2, 1, 3) for i in range(3))         # (B, H, N, d) each


def attn_fwd(qkv, B, N, H, d, scale, drop=None):
    assert drop is None
    q, k, v = _heads(qkv, B, N, H, d)
    out, _ = O.attention_core_fwd(q, k, v, scale)
    o = torch.empty((B * N, H * d), dtype=qkv.dtype)
    o.copy_(out.permute(0, 2, 1, 3).reshape(B * N, H * d))
    return o, torch.zeros((B, H, N), dtype=F32)                       # (the doubles recompute P; lse is a placeholder)


def attn_bwd(qkv, o, do, saved, B, N, H, d, scale, drop=None):
    q, k, v = _heads(qkv, B, N, H, d)
    dq, dk, dv = O.attention_core_bwd(do.reshape(B, N, H, d).float().permute(0, 2, 1, 3), q, k, v, scale)
    dqkv = torch.empty((B * N, 3 * H * d), dtype=qkv.dtype)
    dqkv.view(B, N, 3, H, d).copy_(torch.stack([t.permute(0, 2, 1, 3) for t in (dq, dk, dv)], dim=2))
    return dqkv


def fp8_amax_scale(x, scale2):
    a = x.float().abs().max().clamp_min(1e-12)
    scale2[0] = 448.0 / a; scale2[1] = a / 448.0


def quantize_fp8(x, out, scale_dev=None, scale=1.0):
    out.view(x.shape).copy_(_q8(x, float(scale_dev[0]) if scale_dev is not None else scale, E4M3))


def quantize_fp8_delayed(x, out8, scale2, amax64, fmt):
    CALLS.append(("quantize_fp8_delayed", (tuple(x.shape), out8 is not None, amax64 is not None, fmt)))
    assert out8 is not None or amax64 is not None
    if out8 is not None:
        assert float(scale2[0]) > 0.0, "quantising under a scale that was never decided"
        out8.view(x.shape).copy_(_q8(x, float(scale2[0]), E4M3 if fmt == K.FMT_E4M3 else E5M2))
    if amax64 is not None:
        _rec_amax(amax64, x.float())


def fp8_update_scales_fmt(amax64, scales2, nslots, fmax):
    for s in range(nslots):
        m = float(amax64[s].view(F32).max())
        amax64[s].zero_()
        if m > 0.0:
            fm = float(fmax[s]) if fmax is not None else 448.0
            scales2[s, 0] = fm / m; scales2[s, 1] = m / fm


def fp8_update_scales(amax64, scales2, nslots):
    fp8_update_scales_fmt(amax64, scales2, nslots, None)


def colsum_partials(partials, nparts, ld, cols, out, accumulate=False):
    r = partials[:nparts * ld].view(nparts, ld)[:, :cols].sum(0)
    out.copy_(r + (out.float() if accumulate else 0))


def fold_many(jobs):
    """K.fold_many: the queued folds of a fused backward (ops.deferred_folds), one job at a time."""
    for part, nparts, ld, cols, out, acc in jobs:
        colsum_partials(part, nparts, ld, cols, out, acc)


def colsum(x, rows, cols, ld, out, ws, accumulate=False):
    r = x.reshape(rows, ld)[:, :cols].float().sum(0)
    o = out.view(-1)                        # (cols,) however the caller shapes it ((N, D) for the positional-table gradient)
    o.copy_(r + (o.float() if accumulate else 0))


class _MatRef:
    """What the double of K.mat() hands to the double of K.gemm_generic: the tensor itself + the element strides of vitk_mat."""

    def __init__(self, t, s_row, s_col, s_b1, s_b2, offset):
        self.t, self.s_row, self.s_col, self.s_b1, self.s_b2, self.offset = t, s_row, s_col, s_b1, s_b2, offset
        self._dtype = t.dtype

    def view(self, rows, cols, nb1, nb2):
        return torch.as_strided(self.t, (nb1, nb2, rows, cols), (self.s_b1, self.s_b2, self.s_row, self.s_col), self.t.storage_offset() + self.offset)


def mat(t, s_row, s_col, s_b1=0, s_b2=0, offset=0):
    return _MatRef(t, s_row, s_col, s_b1, s_b2, offset)


def gemm_generic(A, B, Cm, M, N, Kd, nb1=1, nb2=1, bias=None, alpha=1.0, beta=0.0):
    """C[b1, b2] = alpha A[b1, b2] (M, K) . B[b1, b2] (K, N) + beta C (+ bias[n]), every operand by element strides (vitk_gemm_generic)."""
    CALLS.append(("gemm_generic", (M, N, Kd, nb1, nb2)))
    c = Cm.view(M, N, nb1, nb2)
    r = alpha * (A.view(M, Kd, nb1, nb2).float() @ B.view(Kd, N, nb1, nb2).float())
    if beta != 0.0:
        r = r + beta * c.float()
    if bias is not None:
        r = r + bias.float()
    c.copy_(r)


def _thd(t, ld, T, H, d, off=0):
    """(T, H, d) view of per-token rows that are `ld` elements apart, starting `off` elements into the tensor."""
    return torch.as_strided(t, (T, H, d), (ld, d, 1), t.storage_offset() + off)


def rmsnorm_heads_fwd(x, ldx, gamma, y, ldy, rnorm, T, H, d, x_off=0):
    """y = x / max(||x||, 1e-12) * sqrt(d) * gamma[h] per (token, head) (na_vit.py:93-101); rnorm keeps 1 / max(||x||, 1e-12)."""
    xv = _thd(x, ldx, T, H, d, x_off).float()
    r = 1.0 / xv.norm(dim=-1).clamp_min(1e-12)
    _thd(y, ldy, T, H, d).copy_(xv * r.unsqueeze(-1) * (d ** 0.5) * gamma.reshape(1, H, d).float())
    rnorm.view(T, H).copy_(r)


def rmsnorm_heads_bwd(dy, lddy, x, ldx, gamma, rnorm, dx, lddx, dgamma, partials, T, H, d, x_off=0, dx_off=0):
    xv = _thd(x, ldx, T, H, d, x_off).float()
    g = _thd(dy, lddy, T, H, d).float()
    r = rnorm.view(T, H, 1)
    s = d ** 0.5
    gg = g * gamma.reshape(1, H, d).float()
    _thd(dx, lddx, T, H, d, dx_off).copy_(s * r * gg - s * r ** 3 * xv * (xv * gg).sum(-1, keepdim=True))
    dgamma.view(H, d).copy_((g * xv * r * s).sum(0))


def softmax_fwd(sc, p, rows, cols, scale):
    p.view(rows, cols).copy_(O.softmax_fwd(sc.reshape(rows, cols).float() * scale))


def softmax_bwd(p, dp, ds, rows, cols, scale):
    """ds = scale * p * (dp - rowsum(dp * p))  (may run in place on dp)."""
    pv, dv = p.reshape(rows, cols).float(), dp.reshape(rows, cols).float()
    ds.view(rows, cols).copy_(scale * pv * (dv - (dv * pv).sum(-1, keepdim=True)))


def patchify_cpp(img, out, C, H, W, p, row0, ld):
    """NaViT patch extraction of one image, 'c (h p1) (w p2) -> (h w) (c p1 p2)' (na_vit.py:300), into rows row0.. of a (T, ld) matrix."""
    h, w = H // p, W // p
    _rows(out, out.numel() // ld, C * p * p, ld)[row0:row0 + h * w] = img.reshape(C, h, p, w, p).permute(1, 3, 0, 2, 4).reshape(h * w, C * p * p)


def unpatchify_cpp(dpatch, dimg, C, H, W, p, row0, ld):
    """The adjoint of patchify_cpp: rows row0.. of a (T, ld) matrix back into a (C, H, W) image."""
    h, w = H // p, W // p
    rows = _rows(dpatch, dpatch.numel() // ld, C * p * p, ld)[row0:row0 + h * w]
    dimg.reshape(C, H, W).copy_(rows.reshape(h, w, C, p, p).permute(2, 0, 3, 1, 4).reshape(C, H, W))


def gather_add2(x, A, ia, B, ib, out, T, D):
    out.view(T, D).copy_(x.reshape(T, D).float() + A.reshape(-1, D)[ia.long()].float() + B.reshape(-1, D)[ib.long()].float())


def csr_rowsum(g, ptr, rows, out, nseg, D):
    """out[i] = sum of g[rows[j]] for j in [ptr[i], ptr[i + 1]) (deterministic gradient of the embedding gathers)."""
    gv = g.reshape(-1, D).float()
    o = out.view(-1, D)
    for i in range(nseg):
        a, b = int(ptr[i]), int(ptr[i + 1])
        o[i] = gv[rows[a:b].long()].sum(0) if b > a else 0


class _HndRef:
    """Double of K.hnd(): a (tokens, heads, d) view -- element (n, h, :) at offset + n * s_n + h * s_h -- that keeps its tensor."""

    def __init__(self, t, s_h, s_n, offset):
        self.t, self.s_h, self.s_n, self.offset = t, s_h, s_n, offset
        self._dtype = t.dtype

    def view(self, T, H, d):
        return torch.as_strided(self.t, (T, H, d), (self.s_n, self.s_h, 1), self.t.storage_offset() + self.offset)


def hnd(t, s_h, s_n, offset=0):
    return _HndRef(t, s_h, s_n, offset)


def _segments(cu_q, cu_k):
    cq, ck = [int(v) for v in cu_q], [int(v) for v in cu_k]
    return [(cq[i], cq[i + 1], ck[i], ck[i + 1]) for i in range(len(cq) - 1)]


def attn_varlen_fwd_bf16(q, k, v, o, lse, cu_q, cu_k, blk_seg, blk_r0, nblk, tq_total, H, d, scale, drop_p=0.0, drop_seed=0):
    """softmax(scale q k^T) v independently inside every segment (NaViT's masked attention, na_vit.py:161-166, as per-image ranges)."""
    assert drop_p == 0.0
    Tk = int(cu_k[-1])
    qv, kv_, vv, ov = q.view(tq_total, H, d), k.view(Tk, H, d), v.view(Tk, H, d), o.view(tq_total, H, d)
    for q0, q1, k0, k1 in _segments(cu_q, cu_k):
        if q1 == q0:
            continue
        qs, ks, vs = (t.float().permute(1, 0, 2) for t in (qv[q0:q1], kv_[k0:k1], vv[k0:k1]))
        sc = qs @ ks.transpose(-1, -2) * scale
        ov[q0:q1] = (torch.softmax(sc, -1) @ vs).permute(1, 0, 2).to(ov.dtype)
        lse.view(H, tq_total)[:, q0:q1] = torch.logsumexp(sc, -1)


def attn_varlen_bwd_bf16(q, k, v, o, dout, lse, delta, dq, dk, dv, cu_q, cu_k, qblk_seg, qblk_r0, nqblk, kblk_seg, kblk_r0, nkblk, tq_total, H,
                         d, scale, drop_p=0.0, drop_seed=0):
    assert drop_p == 0.0
    Tk = int(cu_k[-1])
    qv, kv_, vv, dov = q.view(tq_total, H, d), k.view(Tk, H, d), v.view(Tk, H, d), dout.view(tq_total, H, d)
    dqv, dkv_, dvv = dq.view(tq_total, H, d), dk.view(Tk, H, d), dv.view(Tk, H, d)
    dkv_.zero_(); dvv.zero_(); dqv.zero_()
    for q0, q1, k0, k1 in _segments(cu_q, cu_k):
        if q1 == q0:
            continue
        qs, ks, vs, ds = (t.float().permute(1, 0, 2) for t in (qv[q0:q1], kv_[k0:k1], vv[k0:k1], dov[q0:q1]))
        gq, gk, gv = O.attention_core_bwd(ds, qs, ks, vs, scale)
        dqv[q0:q1] = gq.permute(1, 0, 2).to(dqv.dtype)
        dkv_[k0:k1] = gk.permute(1, 0, 2).to(dkv_.dtype)
        dvv[k0:k1] = gv.permute(1, 0, 2).to(dvv.dtype)


def concat_tokens(x, front, pos, out, B, Np, F, D):
    """out[b, i] = (i < F ? front[i] : x[b, i - F]) + (pos ? pos[i] : 0): torch.cat((tokens, x), dim=1) + pos[:N] (vit.py:122-127)."""
    behind, F = F < 0, abs(F)       # F < 0: the extra tokens follow x (simple_vit_with_register_tokens.py:113-115)
    o = out.view(B, Np + F, D)
    if behind:
        o[:, :Np] = x.reshape(B, Np, D)
        if F:
            o[:, Np:] = front.reshape(1, F, D).to(o.dtype)
    else:
        if F:
            o[:, :F] = front.reshape(1, F, D).to(o.dtype)
        o[:, F:] = x.reshape(B, Np, D)
    if pos is not None:
        o.copy_(o.float() + pos.reshape(-1, D)[:Np + F].float())


def patchify(img, out, B, C, H, W, p1, p2):
    """rearrange 'b c (h p1) (w p2) -> (b h w) (p1 p2 c)' (vit.py:100)."""
    h, w = H // p1, W // p2
    out.view(B * h * w, p1 * p2 * C).copy_(img.reshape(B, C, h, p1, w, p2).permute(0, 2, 4, 3, 5, 1).reshape(B * h * w, p1 * p2 * C))


def unpatchify(dpatch, dimg, B, C, H, W, p1, p2):
    """gradient of patchify with respect to the image (every pixel belongs to one patch element)."""
    h, w = H // p1, W // p2
    dimg.view(B, C, H, W).copy_(dpatch.reshape(B, h, w, p1, p2, C).permute(0, 5, 1, 3, 2, 4).reshape(B, C, H, W))


def patch_ln_fwd(img, w, b, y, mean, rstd, B, C, H, W, p1, p2, eps=1e-5):
    """vitk_patch_ln_fwd: rearrange (vit.py:100) + LayerNorm(patch_dim) (vit.py:101) in one call, mean / rstd kept."""
    h, ww = H // p1, W // p2
    P = p1 * p2 * C
    x = img.reshape(B, C, h, p1, ww, p2).permute(0, 2, 4, 3, 5, 1).reshape(B * h * ww, P).float()
    m = x.mean(-1); r = 1.0 / torch.sqrt(x.var(-1, unbiased=False) + eps)
    o = (x - m[:, None]) * r[:, None] * w.float()
    if b is not None:
        o = o + b.float()
    y.view(B * h * ww, P).copy_(o)
    mean.copy_(m); rstd.copy_(r)
    CALLS.append(("patch_ln_fwd", (B, C, H, W, p1, p2)))


def patch_ln_bwd_params(dy, img, mean, rstd, partials, B, C, H, W, p1, p2):
    """vitk_patch_ln_bwd_params: dgamma / dbeta partial rows of that LayerNorm, xhat re-formed from the image."""
    h, ww = H // p1, W // p2
    P = p1 * p2 * C
    rows = B * h * ww
    x = img.reshape(B, C, h, p1, ww, p2).permute(0, 2, 4, 3, 5, 1).reshape(rows, P).float()
    xh = (x - mean[:, None]) * rstd[:, None]
    g = dy.reshape(rows, P).float()
    nblk = partials.numel() // (2 * P)
    pv = partials.view(2, nblk, P)
    pv.zero_()
    pv[0, 0] = (g * xh).sum(0); pv[1, 0] = g.sum(0)
    CALLS.append(("patch_ln_bwd_params", (B, C, H, W, p1, p2)))


def copy_cols(src, ld_src, dst, ld_dst, rows, cols_copy, cols_dst):
    d = _rows(dst, rows, cols_dst, ld_dst)
    d.zero_()
    d[:, :cols_copy] = _rows(src, rows, cols_copy, ld_src)


def write_cls_rows(x, cls, pos, B, N, D, ncls):
    x.view(B, N, D)[:, :ncls] = (cls.float() + pos.reshape(-1, D)[:ncls].float()).to(x.dtype)


def mean_pool_fwd(x, out, B, N, D):
    out.view(B, D).copy_(x.reshape(B, N, D).float().mean(1))


def mean_pool_bwd(dout, dx, B, N, D):
    dx.view(B, N, D).copy_((dout.reshape(B, 1, D).float() / N).expand(B, N, D))


def transpose(x, out, rows, cols):
    out.view(cols, rows).copy_(x.reshape(rows, cols).t())


def add_rows(a, b, bias, out, rows, cols):
    r = a.reshape(rows, cols).float() + b.reshape(rows, cols).float()
    out.view(rows, cols).copy_(r + (bias.float() if bias is not None else 0))


def cast(x, y):
    y.copy_(x)


def cast_many(srcs, dsts):
    for s, d in zip(srcs, dsts):
        d.copy_(s)


def gelu_fwd(x, y):
    y.copy_(O.gelu_fwd(x.float()))


def gelu_bwd(dy, x, dx):
    dx.copy_(O.gelu_bwd(dy.float(), x.float()))
def gemm_nt_bf16_mul_aux8_colsum(A, lda, W, ldw, C, ldc, M, N, Kd, aux8, partials):
    CALLS.append(("gemm_nt_bf16_mul_aux8_colsum", (M, N, Kd)))
    _epilogue(A.reshape(M, Kd).float() @ _w_plain(W, ldw, N, Kd).t(), C, M, N, L.EPI_MUL_AUX8, None, None, aux8, partials)


def require_device(*ts):
    return None


_K_DOUBLES = dict(gemm_nt_bf16=gemm_nt_bf16, gemm_nt_bf16_gelu_bwd_colsum=gemm_nt_bf16_gelu_bwd_colsum, gemm_nt_bf16_mul_aux_colsum=gemm_nt_bf16_mul_aux_colsum, gemm_nt_bf16_mul_aux8_colsum=gemm_nt_bf16_mul_aux8_colsum, gemm_nt_fp8_v2=gemm_nt_fp8_v2,
                  gemm_nt_fp8_ex=gemm_nt_fp8_ex, pack_w_nt=pack_w_nt, pack_w_nt_many=pack_w_nt_many, gemm_tn_bf16=gemm_tn_bf16, gemm_tn_bf16_pair=gemm_tn_bf16_pair, gemm_tn_fp8=gemm_tn_fp8, layernorm_fwd=layernorm_fwd,
                  fp8_amax_scale=fp8_amax_scale, quantize_fp8=quantize_fp8, quantize_fp8_delayed=quantize_fp8_delayed,
                  fp8_update_scales_fmt=fp8_update_scales_fmt, fp8_update_scales=fp8_update_scales, colsum_partials=colsum_partials, fold_many=fold_many,
                  colsum=colsum, transpose=transpose, add_rows=add_rows, cast=cast, cast_many=cast_many, gelu_fwd=gelu_fwd, gelu_bwd=gelu_bwd, patchify=patchify, unpatchify=unpatchify, patch_ln_fwd=patch_ln_fwd, patch_ln_bwd_params=patch_ln_bwd_params,
                  copy_cols=copy_cols, write_cls_rows=write_cls_rows, mat=mat, gemm_generic=gemm_generic, concat_tokens=concat_tokens, hnd=hnd, attn_varlen_fwd_bf16=attn_varlen_fwd_bf16, attn_varlen_bwd_bf16=attn_varlen_bwd_bf16,
                  patchify_cpp=patchify_cpp, unpatchify_cpp=unpatchify_cpp, gather_add2=gather_add2, csr_rowsum=csr_rowsum,
                  rmsnorm_heads_fwd=rmsnorm_heads_fwd, rmsnorm_heads_bwd=rmsnorm_heads_bwd,
                  softmax_fwd=softmax_fwd, softmax_bwd=softmax_bwd, mean_pool_fwd=mean_pool_fwd, mean_pool_bwd=mean_pool_bwd,
                  require_device=require_device)
_OPS_DOUBLES = dict(attn_fwd=attn_fwd, attn_bwd=attn_bwd, ln_bwd=ln_bwd)


@contextlib.contextmanager
def installed():
    """Swap the doubles in (and torch.cuda.is_current_stream_capturing, which raises without a device), restore on exit."""
    saved_k = {n: getattr(K, n) for n in _K_DOUBLES}
    saved_o = {n: getattr(ops, n) for n in _OPS_DOUBLES}
    saved_cap = torch.cuda.is_current_stream_capturing
    del CALLS[:]
    try:
        for n, f in _K_DOUBLES.items():
            setattr(K, n, f)
        for n, f in _OPS_DOUBLES.items():
            setattr(ops, n, f)
        torch.cuda.is_current_stream_capturing = lambda: False
        yield CALLS
    finally:
        for n, f in saved_k.items():
            setattr(K, n, f)
        for n, f in saved_o.items():
            setattr(ops, n, f)
        torch.cuda.is_current_stream_capturing = saved_cap
        ops._WT_CACHE.clear(); ops._WP_CACHE.clear()
