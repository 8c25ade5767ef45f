"""Drop-in for vit_pytorch/na_vit.py (NaViT: group_images_by_max_seq_len :38-77, LayerNorm :82-89,
RMSNorm :93-101, FeedForward :105-113, Attention :115-169, Transformer :171-193, NaViT :195-402) on
libvitk kernels.

MI355X-first data layout.  The reference pads every pack of images to (b, n, d) and drives
F.scaled_dot_product_attention with a dense boolean (b, 1, n, n) mask "same image and key is not padding"
(:335-337), pads positions and image ids (:335-342) and selects the real images at the end with a boolean
index (:393-396).  Here the tokens of ALL images of ALL packs are kept unpadded in one (T, dim) matrix; the
only thing attention needs is the list of per-image token ranges, because the mask is block diagonal with
one block per image: attention is launched per (image, 128-query block, head) (vitk_attn_varlen_*).  The
attention pool (one learned query per image, :371-387) is the same kernel with one query row per image.
Results for real tokens are identical to the padded formulation (padded rows never reach an output); packing
into groups therefore changes nothing numerically and `group_images` is honoured only as an API.

Same constructor, state_dict keys/shapes (gamma/beta LayerNorms, to_q / to_kv, q_norm / k_norm,
pos_embed_height / pos_embed_width, attn_pool_queries, attn_pool.*, mlp_head.{0,1}) and forward() contract
(list of images or list of lists -> (total_images, num_classes)).
"""
from __future__ import annotations

from typing import List

import numpy as np
import torch
from torch import nn, Tensor

from ._epoch import note_grad_mode
from . import functional as Fn
from . import kernels as K
from . import ops
from ._lib import VitkError

F32 = torch.float32


def pair(t):
    return t if isinstance(t, tuple) else (t, t)


def _token_count(image: Tensor, patch_size: int, drop_fraction) -> int:
    """Tokens an image contributes to a pack: its patch grid, shrunk by the token-dropout fraction for its size (floored)."""
    height, width = image.shape[-2:]
    frac = drop_fraction(height, width) if callable(drop_fraction) else (0.0 if drop_fraction is None else float(drop_fraction))
    return int((height // patch_size) * (width // patch_size) * (1 - frac))


def group_images_by_max_seq_len(images: List[Tensor], patch_size: int, calc_token_dropout=None, max_seq_len=2048) -> List[List[Tensor]]:
    """First-fit packing IN INPUT ORDER: an image joins the current group unless that would push the group's token count
    past `max_seq_len`, in which case the group is closed and the image opens the next one (the behaviour of na_vit.py:38-77;
    `calc_token_dropout` may be None, a number, or a callable (height, width) -> fraction)."""
    packs: List[List[Tensor]] = []
    current: List[Tensor] = []
    used = 0
    for image in images:
        if not isinstance(image, Tensor):
            raise AssertionError("group_images_by_max_seq_len expects a list of image tensors")
        need = _token_count(image, patch_size, calc_token_dropout)
        assert need <= max_seq_len, f'image with dimensions {tuple(image.shape[-2:])} exceeds maximum sequence length'
        if used + need > max_seq_len:
            packs.append(current)
            current, used = [], 0
        current.append(image)
        used += need
    if current:
        packs.append(current)
    return packs


from .segments import Segments  # noqa: E402  (re-exported: tests and users import it from here)


class _QKNormAttnFn(torch.autograd.Function):
    """q_norm / k_norm (RMSNorm per head) + scaled_dot_product_attention(scale = 1) over segments
    (na_vit.py:147-168).  q: (Tq, I); kv: (Tk, 2I) = k | v, consumed in place; returns (Tq, I)."""

    @staticmethod
    def forward(ctx, q, kv, gq, gk, segs: Segments, heads: int, drop_p: float = 0.0, drop_seed: int = 0):
        K.require_device(q, kv)
        q = q.contiguous(); kv = kv.contiguous()
        Tq, I = q.shape
        Tk = kv.shape[0]
        d = I // heads
        T = q.dtype
        if d % 4 or d > 256:
            raise VitkError(f"q / k RMSNorm attention (NaViT, simple_vit_with_qk_norm) needs dim_head % 4 == 0 and dim_head <= 256 (got {d}): vitk_rmsnorm_heads_* are written for 16-byte lanes")
        flash = ops.attn_varlen_ok(T, d)       # 16-bit, dim_head 32 / 48 / 64 / 80 / 96; everything else: per-segment materialising path
        qn = torch.empty_like(q); kn = torch.empty((Tk, I), dtype=T, device=q.device)
        rq = torch.empty(Tq * heads, dtype=F32, device=q.device); rk = torch.empty(Tk * heads, dtype=F32, device=q.device)
        gqf, gkf = gq.reshape(heads, d).contiguous(), gk.reshape(heads, d).contiguous()
        K.rmsnorm_heads_fwd(q, I, gqf, qn, I, rq, Tq, heads, d)
        K.rmsnorm_heads_fwd(kv, 2 * I, gkf, kn, I, rk, Tk, heads, d)
        o = torch.empty((Tq, I), dtype=T, device=q.device)
        if flash:
            lse = torch.empty((heads, Tq), dtype=F32, device=q.device)
            K.attn_varlen_fwd_bf16(K.hnd(qn, d, I), K.hnd(kn, d, I), K.hnd(kv, d, 2 * I, offset=I), K.hnd(o, d, I), lse,
                                   segs.cu_q, segs.cu_k, segs.qblk_seg, segs.qblk_r0, segs.nqblk, Tq, heads, d, 1.0, drop_p, drop_seed)
            saved = lse
        else:  # f32 validation mode (and head widths the flash kernels do not take): per-segment materialising path on the coverage kernels
            if drop_p > 0.0:
                raise VitkError("attention dropout of NaViT is fused into the 16-bit attention kernels only (bfloat16 / float16 with dim_head 32 / 48 / 64 / 80 / 96, or eval())")
            saved = []
            for s in range(segs.nseg):
                q0, nq = int(segs.cu_q_host[s]), segs.q_lens[s]
                k0, nk = int(segs.cu_k_host[s]), segs.k_lens[s]
                S = torch.empty((heads, nq, nk), dtype=T, device=q.device)
                K.gemm_generic(K.mat(qn, I, 1, 0, d, offset=q0 * I), K.mat(kn, 1, I, 0, d, offset=k0 * I),
                               K.mat(S, nk, 1, 0, nq * nk), nq, nk, d, nb1=1, nb2=heads)
                P = torch.empty_like(S)
                K.softmax_fwd(S, P, heads * nq, nk, 1.0)
                K.gemm_generic(K.mat(P, nk, 1, 0, nq * nk), K.mat(kv, 2 * I, 1, 0, d, offset=k0 * 2 * I + I),
                               K.mat(o, I, 1, 0, d, offset=q0 * I), nq, d, nk, nb1=1, nb2=heads)
                saved.append(P)
        ctx.save_for_backward(q, kv, gqf, gkf, qn, kn, o, rq, rk)
        ctx.att = saved
        ctx.meta = (segs, heads, d, gq.shape, gk.shape)
        ctx.flash = flash
        ctx.drop = (drop_p, drop_seed)
        return o

    @staticmethod
    def backward(ctx, do):
        q, kv, gqf, gkf, qn, kn, o, rq, rk = ctx.saved_tensors
        segs, heads, d, gq_shape, gk_shape = ctx.meta
        Tq, I = q.shape
        Tk = kv.shape[0]
        T = q.dtype
        do = Fn._to(do, T)
        dqn = torch.empty_like(qn); dkn = torch.empty_like(kn)
        dkv = torch.empty_like(kv)
        if ctx.flash:
            delta = torch.empty((heads, Tq), dtype=F32, device=q.device)
            K.attn_varlen_bwd_bf16(K.hnd(qn, d, I), K.hnd(kn, d, I), K.hnd(kv, d, 2 * I, offset=I), K.hnd(o, d, I),
                                   K.hnd(do, d, I), ctx.att, delta, K.hnd(dqn, d, I), K.hnd(dkn, d, I),
                                   K.hnd(dkv, d, 2 * I, offset=I), segs.cu_q, segs.cu_k, segs.qblk_seg, segs.qblk_r0,
                                   segs.nqblk, segs.kblk_seg, segs.kblk_r0, segs.nkblk, Tq, heads, d, 1.0, *ctx.drop)
        else:
            for s in range(segs.nseg):
                q0, nq = int(segs.cu_q_host[s]), segs.q_lens[s]
                k0, nk = int(segs.cu_k_host[s]), segs.k_lens[s]
                P = ctx.att[s]
                pm = K.mat(P, nk, 1, 0, nq * nk); pmT = K.mat(P, 1, nk, 0, nq * nk)
                dom = K.mat(do, I, 1, 0, d, offset=q0 * I)
                K.gemm_generic(pmT, dom, K.mat(dkv, 2 * I, 1, 0, d, offset=k0 * 2 * I + I), nk, d, nq, nb1=1, nb2=heads)      # dV
                dP = torch.empty_like(P)
                K.gemm_generic(dom, K.mat(kv, 1, 2 * I, 0, d, offset=k0 * 2 * I + I), K.mat(dP, nk, 1, 0, nq * nk), nq, nk, d, nb1=1, nb2=heads)
                K.softmax_bwd(P, dP, dP, heads * nq, nk, 1.0)
                K.gemm_generic(K.mat(dP, nk, 1, 0, nq * nk), K.mat(kn, I, 1, 0, d, offset=k0 * I), K.mat(dqn, I, 1, 0, d, offset=q0 * I), nq, d, nk, nb1=1, nb2=heads)
                K.gemm_generic(K.mat(dP, 1, nk, 0, nq * nk), K.mat(qn, I, 1, 0, d, offset=q0 * I), K.mat(dkn, I, 1, 0, d, offset=k0 * I), nk, d, nq, nb1=1, nb2=heads)
        dq = torch.empty_like(q)
        dgq = torch.empty_like(gqf); dgk = torch.empty_like(gkf)
        pq = torch.empty(K.rmsnorm_heads_partials(Tq, heads, d), dtype=F32, device=q.device)
        pk = torch.empty(K.rmsnorm_heads_partials(Tk, heads, d), dtype=F32, device=q.device)
        K.rmsnorm_heads_bwd(dqn, I, q, I, gqf, rq, dq, I, dgq, pq, Tq, heads, d)
        K.rmsnorm_heads_bwd(dkn, I, kv, 2 * I, gkf, rk, dkv, 2 * I, dgk, pk, Tk, heads, d)
        return dq, dkv, dgq.view(gq_shape), dgk.view(gk_shape), None, None, None, None


class _PackPatchesFn(torch.autograd.Function):
    """The packed (T, C p p) patch matrix of a list of images -- `rearrange(image, 'c (h p1) (w p2) -> (h w) (c p1 p2)')` per image
    (na_vit.py:300), token dropout's `patches[keep]` (:306-314) and the concatenation, written straight into one matrix -- as ONE autograd
    node over the images, so that `image.requires_grad_()` gets its gradient like under the reference (round 6; it used to raise): the
    backward scatters the kept rows back (dropped patches: zero) and un-patchifies per image (vitk_unpatchify_cpp)."""

    @staticmethod
    def forward(ctx, meta, *images):
        c, p, keeps, lens, dtype, device = meta
        P = c * p * p
        T = int(sum(lens))
        patches = torch.empty((T, P), dtype=dtype, device=device)
        row0 = 0
        for image, keep, n in zip(images, keeps, lens):
            image = image.contiguous()
            ih, iw = image.shape[-2:]
            if keep is None:
                K.patchify_cpp(image, patches, c, ih, iw, p, row0, P)
            else:
                full = torch.empty(((ih // p) * (iw // p), P), dtype=dtype, device=device)
                K.patchify_cpp(image, full, c, ih, iw, p, 0, P)
                ptr = torch.arange(n + 1, dtype=torch.int32, device=device)
                K.csr_rowsum(full, ptr, keep.to(torch.int32), patches[row0:row0 + n], n, P)   # row gather
            row0 += n
        ctx.meta = meta
        ctx.shapes = [tuple(im.shape) for im in images]
        return patches

    @staticmethod
    def backward(ctx, g):
        c, p, keeps, lens, dtype, device = ctx.meta
        P = c * p * p
        g = g.contiguous()
        grads = []
        row0 = 0
        for i, (shape, keep, n) in enumerate(zip(ctx.shapes, keeps, lens)):
            if not ctx.needs_input_grad[1 + i]:
                grads.append(None); row0 += n
                continue
            ih, iw = shape[-2:]
            dimg = torch.empty(shape, dtype=g.dtype, device=g.device)
            if keep is None:
                K.unpatchify_cpp(g, dimg, c, ih, iw, p, row0, P)
            else:       # the dropped patches' rows are zero
                full = torch.zeros((1, (ih // p) * (iw // p), P), dtype=g.dtype, device=g.device)
                K.gather_tokens(g[row0:row0 + n].contiguous().view(1, n, P), keep.to(torch.int32).view(1, n), full, 1, (ih // p) * (iw // p), n, P, scatter=True)
                K.unpatchify_cpp(full, dimg, c, ih, iw, p, 0, P)
            grads.append(dimg)
            row0 += n
        return (None, *grads)


class _PosEmbedFn(torch.autograd.Function):
    """x + pos_embed_height[h_idx] + pos_embed_width[w_idx] (na_vit.py:354-359); deterministic backward
    through CSR lists of the tokens that use each table row."""

    @staticmethod
    def forward(ctx, x, ph, pw, h_idx, w_idx, h_csr, w_csr):
        K.require_device(x)
        x = x.contiguous()
        T, D = x.shape
        out = torch.empty_like(x)
        K.gather_add2(x, ph, h_idx, pw, w_idx, out, T, D)
        ctx.csr = (h_csr, w_csr, ph.shape, pw.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        h_csr, w_csr, ph_shape, pw_shape = ctx.csr
        g = g.contiguous()
        D = g.shape[1]
        dph = torch.empty(ph_shape, dtype=g.dtype, device=g.device)
        dpw = torch.empty(pw_shape, dtype=g.dtype, device=g.device)
        K.csr_rowsum(g, h_csr[0], h_csr[1], dph, ph_shape[0], D)
        K.csr_rowsum(g, w_csr[0], w_csr[1], dpw, pw_shape[0], D)
        return g, dph, dpw, None, None, None, None


class _ExpandRowsFn(torch.autograd.Function):
    """repeat(vec, 'd -> n d') (na_vit.py:373); backward = column sum."""

    @staticmethod
    def forward(ctx, vec, n: int, ptr, rows):
        D = vec.shape[0]
        out = torch.empty((n, D), dtype=vec.dtype, device=vec.device)
        K.csr_rowsum(vec.contiguous(), ptr, rows, out, n, D)
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        D = g.shape[1]
        dv = torch.empty(D, dtype=g.dtype, device=g.device)
        ops.colsum(g, ctx.n, D, dv)
        return dv, None, None, None


# ---- modules (same state_dict as the reference) -----------------------------------------------------
class LayerNorm(nn.Module):
    """LayerNorm without a learned bias (na_vit.py:82-89): `gamma` parameter, zero `beta` buffer."""

    def __init__(self, dim):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(dim))
        self.register_buffer('beta', torch.zeros(dim))

    def forward(self, x):
        return Fn.LayerNormFn.apply(x, self.gamma, None)


class RMSNorm(nn.Module):
    """q/k normalisation (na_vit.py:93-101); applied inside _QKNormAttnFn."""

    def __init__(self, heads, dim):
        super().__init__()
        self.scale = dim ** 0.5
        self.gamma = nn.Parameter(torch.ones(heads, 1, dim))


def FeedForward(dim, hidden_dim, dropout=0.):
    return nn.Sequential(
        LayerNorm(dim),
        Fn.Linear(dim, hidden_dim),
        Fn.GELU(),
        Fn.Dropout(dropout),
        Fn.Linear(hidden_dim, dim),
        Fn.Dropout(dropout),
    )


class Attention(nn.Module):
    _instances = [0]       # construction index: per-instance dropout seed sequences (see vit.Transformer)

    def __init__(self, dim, heads=8, dim_head=64, dropout=0.):
        super().__init__()
        inner_dim = dim_head * heads
        self._drop_calls, self._drop_salt = 0, Attention._instances[0]
        Attention._instances[0] += 1
        self.heads = heads
        self.norm = LayerNorm(dim)
        self.q_norm = RMSNorm(heads, dim_head)
        self.k_norm = RMSNorm(heads, dim_head)
        self.dropout_p = dropout
        self.to_q = Fn.Linear(dim, inner_dim, bias=False)
        self.to_kv = Fn.Linear(dim, inner_dim * 2, bias=False)
        self.to_out = nn.Sequential(
            Fn.Linear(inner_dim, dim, bias=False),
            Fn.Dropout(dropout),
        )

    def _drop_state(self):
        """(calls so far, per-instance salt) of the fused-dropout seed sequence.  Plain instance state: modules unpickled from an older
        torch.save(model) lack it (defaults here), and copy.deepcopy re-salts (__deepcopy__) so that an EMA / teacher copy does not
        draw the masks of its source."""
        if "_drop_salt" not in self.__dict__:
            self.__dict__["_drop_calls"] = 0
            self.__dict__["_drop_salt"] = Attention._instances[0]
            Attention._instances[0] += 1
        return self.__dict__["_drop_calls"], self.__dict__["_drop_salt"]

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = copy.deepcopy(v, memo)
        new.__dict__["_drop_calls"] = 0
        new.__dict__["_drop_salt"] = Attention._instances[0]
        Attention._instances[0] += 1
        return new

    def forward(self, x, segs: Segments, context=None):
        """x: (Tq, dim) packed query-side tokens; context: (Tk, dim) packed key-side tokens (default: x)."""
        x = self.norm(x)
        kv_input = x if context is None else context
        q = self.to_q(x)
        kv = self.to_kv(kv_input)
        p, seed = 0.0, 0
        if self.training and self.dropout_p > 0.:       # F.scaled_dot_product_attention(dropout_p=...) of na_vit.py:163, in-kernel
            p = float(self.dropout_p)
            seed = (int(torch.initial_seed()) + 0x9E3779B1 * self._drop_state()[0] + 0x85EBCA6B * self._drop_state()[1] + 0x632BE5AB * Fn.dist_rank()) & 0xffffffff
            self.__dict__["_drop_calls"] = self._drop_state()[0] + 1
        out = _QKNormAttnFn.apply(q, kv, self.q_norm.gamma, self.k_norm.gamma, segs, self.heads, p, seed)
        return self.to_out(out)


class Transformer(nn.Module):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout=0.):
        super().__init__()
        self.layers = nn.ModuleList([])
        for _ in range(depth):
            self.layers.append(nn.ModuleList([
                Attention(dim, heads=heads, dim_head=dim_head, dropout=dropout),
                FeedForward(dim, mlp_dim, dropout=dropout),
            ]))
        self.norm = LayerNorm(dim)

    # fused-dropout seed sequence: per instance, per training call (see vit.Transformer); not in state_dict
    _instances = [0]

    def _drop_state(self):
        if "_drop_salt" not in self.__dict__:
            self.__dict__["_drop_calls"] = 0
            self.__dict__["_drop_salt"] = Transformer._instances[0]
            Transformer._instances[0] += 1
        return self.__dict__["_drop_calls"], self.__dict__["_drop_salt"]

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = copy.deepcopy(v, memo)
        new.__dict__["_drop_calls"] = 0
        new.__dict__["_drop_salt"] = Transformer._instances[0]
        Transformer._instances[0] += 1
        return new

    def _dropout_p(self):
        """The common p of the block's active dropouts (0.0 if inactive); None if the modules disagree (a user edited them)."""
        if not self.training:
            return 0.0
        ps = set()
        for attn, ff in self.layers:
            ps.add(float(attn.dropout_p))
            ps.update(float(m.p) for m in list(ff) + list(attn.to_out) if isinstance(m, nn.Dropout))
        return ps.pop() if len(ps) == 1 else (0.0 if not ps else None)

    def _fusable(self, x) -> bool:
        """The fused packed engine covers the block in 16 bit with dim_head 64 when nothing observes its inside; active dropout is
        fused too when the GEMM shapes are those the fused-dropout epilogues serve (engine.packed_dropout_fusable)."""
        if x.dtype not in (torch.bfloat16, torch.float16) or self.norm.gamma.dtype != x.dtype:
            return False
        for pair_ in self.layers:
            if len(pair_) != 2:
                return False
            attn, ff = pair_
            # exactly the blocks the constructor builds (na_vit.py:96-169): model surgery (a LoRA / quantised Linear subclass, an extra
            # module, a foreign block) runs op by op through the modules that are there
            if not (type(attn) is Attention and type(ff) is nn.Sequential and len(ff) == 6 and type(ff[0]) is LayerNorm and Fn.plain_linear(ff[1])
                    and Fn.exactly(ff[2], Fn.GELU) and Fn.exactly(ff[3], Fn.Dropout) and Fn.plain_linear(ff[4]) and Fn.exactly(ff[5], Fn.Dropout)
                    and type(attn.norm) is LayerNorm and type(attn.q_norm) is RMSNorm and type(attn.k_norm) is RMSNorm
                    and Fn.plain_linear(attn.to_q) and Fn.plain_linear(attn.to_kv) and type(attn.to_out) is nn.Sequential and len(attn.to_out) == 2
                    and Fn.plain_linear(attn.to_out[0]) and Fn.exactly(attn.to_out[1], Fn.Dropout)):
                return False
            if not ops.attn_varlen_ok(x.dtype, attn.q_norm.gamma.shape[-1]):       # the flash kernels' head widths: 32 / 48 / 64 / 80 / 96
                return False
            if any(Fn.observed(m) for m in list(attn.modules()) + list(ff.modules())):
                return False
        if Fn.observed(self.norm) or type(self.norm) is not LayerNorm:
            return False
        p = self._dropout_p()
        if p is None:
            return False
        if p > 0. and len(self.layers):
            from . import engine as E
            attn, ff = self.layers[0]
            return E.packed_dropout_fusable(x.dtype, x.shape[0], x.shape[1], attn.heads, attn.q_norm.gamma.shape[-1], ff[1].weight.shape[0])
        return True

    def forward(self, x, segs: Segments):
        if self._fusable(x):
            from . import engine as E
            params = []
            for attn, ff in self.layers:
                params += E.pack_navit_layer_params(attn, ff)
            heads = self.layers[0][0].heads if len(self.layers) else 1
            dim_head = self.layers[0][0].q_norm.gamma.shape[-1] if len(self.layers) else 64
            p = self._dropout_p()
            seed = 0
            if p > 0.:
                seed = (int(torch.initial_seed()) + 0x9E3779B1 * self._drop_state()[0] + 0x85EBCA6B * self._drop_state()[1] + 0x632BE5AB * Fn.dist_rank()) & 0xffffffff
                self.__dict__["_drop_calls"] = self._drop_state()[0] + 1
            note_grad_mode(torch.is_grad_enabled())      # Function.forward cannot see no_grad(): it decides what to keep from this
            return E.PackedTransformerFn.apply(x, segs, heads, dim_head, float(p), seed, self.norm.gamma, *params)
        for attn, ff in self.layers:
            x = Fn.AddFn.apply(attn(x, segs), x)
            x = Fn.AddFn.apply(ff(x), x)
        return self.norm(x)


class NaViT(nn.Module):
    def __init__(self, *, image_size, patch_size, num_classes, dim, depth, heads, mlp_dim, channels=3, dim_head=64,
                 dropout=0., emb_dropout=0., token_dropout_prob=None):
        super().__init__()
        image_height, image_width = pair(image_size)

        self.calc_token_dropout = None
        if callable(token_dropout_prob):
            self.calc_token_dropout = token_dropout_prob
        elif isinstance(token_dropout_prob, (float, int)):
            assert 0. <= token_dropout_prob < 1.
            token_dropout_prob = float(token_dropout_prob)
            self.calc_token_dropout = lambda height, width: token_dropout_prob

        assert image_height % patch_size == 0 and image_width % patch_size == 0, 'Image dimensions must be divisible by the patch size.'

        patch_height_dim, patch_width_dim = (image_height // patch_size), (image_width // patch_size)
        patch_dim = channels * (patch_size ** 2)

        self.channels = channels
        self.patch_size = patch_size

        self.to_patch_embedding = nn.Sequential(
            LayerNorm(patch_dim),
            Fn.Linear(patch_dim, dim),
            LayerNorm(dim),
        )

        self.pos_embed_height = nn.Parameter(torch.randn(patch_height_dim, dim))
        self.pos_embed_width = nn.Parameter(torch.randn(patch_width_dim, dim))

        self.dropout = Fn.Dropout(emb_dropout)

        self.transformer = Transformer(dim, depth, heads, dim_head, mlp_dim, dropout)

        self.attn_pool_queries = nn.Parameter(torch.randn(dim))
        self.attn_pool = Attention(dim=dim, dim_head=dim_head, heads=heads)

        self.to_latent = nn.Identity()

        self.mlp_head = nn.Sequential(
            LayerNorm(dim),
            Fn.Linear(dim, num_classes, bias=False),
        )

    @property
    def device(self):
        return next(self.parameters()).device

    @Fn.autocast_aware
    def forward(self, batched_images, group_images=False, group_max_seq_len=2048):
        p, c, device = self.patch_size, self.channels, self.device
        has_token_dropout = self.calc_token_dropout is not None and self.training
        dtype = self.pos_embed_height.dtype

        if group_images:
            batched_images = group_images_by_max_seq_len(batched_images, patch_size=p,
                                                         calc_token_dropout=self.calc_token_dropout if self.training else None,
                                                         max_seq_len=group_max_seq_len)
        if torch.is_tensor(batched_images[0]):
            batched_images = [batched_images]
        images = [img for group in batched_images for img in group]   # pack-major image order == the reference's output order
        K.require_device(*images)

        # ---- host side: token counts, positions, (optional) token dropout ----
        # token dropout (na_vit.py:307-313) draws its kept indices on the GPU; ALL images' draws are queued first and come back in
        # ONE device-to-host copy (a copy per image used to stall the launch queue once per image)
        grids, keeps = [], []
        for image in images:
            assert image.ndim == 3 and image.shape[0] == c
            ih, iw = image.shape[-2:]
            assert ih % p == 0 and iw % p == 0, f'height and width {(ih, iw)} of images must be divisible by patch size {p}'
            ph, pw = ih // p, iw // p
            grids.append((ph, pw))
            keep = None
            if has_token_dropout:
                n = ph * pw
                num_keep = max(1, int(n * (1 - self.calc_token_dropout(ih, iw))))
                keep = torch.randn((n,), device=device).topk(num_keep, dim=-1).indices     # na_vit.py:311
            keeps.append(keep)
        keep_host = torch.cat(keeps).cpu().numpy() if has_token_dropout else None
        lens, h_list, w_list, k0 = [], [], [], 0
        for (ph, pw), keep in zip(grids, keeps):
            hi = np.repeat(np.arange(ph, dtype=np.int32), pw)
            wi = np.tile(np.arange(pw, dtype=np.int32), ph)
            if keep is not None:
                ki = keep_host[k0:k0 + keep.numel()]
                k0 += keep.numel()
                hi, wi = hi[ki], wi[ki]
            lens.append(len(hi)); h_list.append(hi); w_list.append(wi)
        T = int(sum(lens))
        P = c * p * p
        segs = Segments(lens, lens, device)
        h_all, w_all = np.concatenate(h_list), np.concatenate(w_list)

        def csr(idx, n):
            order = np.argsort(idx, kind="stable").astype(np.int32)
            ptr = np.concatenate([[0], np.cumsum(np.bincount(idx, minlength=n))]).astype(np.int32)
            return torch.from_numpy(ptr).to(device), torch.from_numpy(order).to(device)

        h_idx, w_idx = torch.from_numpy(h_all).to(device), torch.from_numpy(w_all).to(device)
        h_csr, w_csr = csr(h_all, self.pos_embed_height.shape[0]), csr(w_all, self.pos_embed_width.shape[0])

        # ---- patches: 'c (h p1) (w p2) -> (h w) (c p1 p2)' written straight into the packed (T, P) matrix (one autograd node over the images) ----
        for image in images:
            if image.dtype != dtype:
                raise VitkError(f"image dtype {image.dtype} != parameter dtype {dtype}")
        patches = _PackPatchesFn.apply((c, p, keeps, lens, dtype, device), *images)

        x = self.to_patch_embedding(patches)
        x = _PosEmbedFn.apply(x, self.pos_embed_height, self.pos_embed_width, h_idx, w_idx, h_csr, w_csr)
        x = self.dropout(x)

        x = self.transformer(x, segs)

        # ---- attention pooling: one learned query per image against that image's tokens ----
        nimg = len(images)
        pool_segs = Segments([1] * nimg, lens, device)
        ptr = torch.arange(nimg + 1, dtype=torch.int32, device=device)
        rows = torch.zeros(nimg, dtype=torch.int32, device=device)
        queries = _ExpandRowsFn.apply(self.attn_pool_queries, nimg, ptr, rows)
        x = Fn.AddFn.apply(self.attn_pool(queries, pool_segs, context=x), queries)

        x = self.to_latent(x)
        return self.mlp_head(x)


Fn.eager_modules(globals())
