"""SimpleViT with query / key normalisation on libvitk kernels -- the module contract of vit_pytorch/simple_vit_with_qk_norm.py.

Differences from SimpleViT (simple_vit_with_qk_norm.py:29-37, 52-84): every head's q and k rows are L2-normalised and multiplied
by sqrt(dim_head) * gamma[head] (`q_norm.gamma`, `k_norm.gamma` of shape (heads, 1, dim_head), initialised to 1 / sqrt(dim_head)),
and the 1 / sqrt(dim_head) score scale is dropped.  The reference's `linear_head` of this file is an `nn.LayerNorm(dim)`
(simple_vit_with_qk_norm.py:128 -- `num_classes` is accepted and unused, the model returns (B, dim)); reproduced as is, since
`state_dict` keys and outputs are the contract.

Kernels: the per-head RMSNorm (`vitk_rmsnorm_heads_*`) and the scale-1 attention are the ones NaViT uses (na_vit._QKNormAttnFn,
chunked flash kernels over one segment per image; dim_head % 4 == 0, <= 256: the per-head RMSNorm kernels take 16-byte lanes); LayerNorm / Linear / GELU / residual
adds are the op-level Functions of functional.py.
"""
from __future__ import annotations

import torch
from torch import nn

from . import functional as Fn
from . import kernels as K
from .na_vit import _QKNormAttnFn
from .segments import uniform_segments
from .simple_vit import FeedForward, posemb_sincos_2d  # noqa: F401  (same FeedForward: LayerNorm, Linear, GELU, Linear)
from .vit import pair


class RMSNorm(nn.Module):
    """normalize(x, dim=-1) * sqrt(dim) * gamma, gamma: (heads, 1, dim) (simple_vit_with_qk_norm.py:29-37).  Applied inside
    `_QKNormAttnFn`; calling the module directly is served by the same kernel."""

    def __init__(self, heads, dim):
        super().__init__()
        self.scale = dim ** 0.5
        self.gamma = nn.Parameter(torch.ones(heads, 1, dim) / self.scale)

    def forward(self, x):           # x: (b, h, n, d)
        b, h, n, d = x.shape
        rows = x.permute(0, 2, 1, 3).reshape(b * n, h * d)
        return _HeadRMSNormFn.apply(rows, self.gamma, h).view(b, n, h, d).permute(0, 2, 1, 3)


class _HeadRMSNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, heads):
        x = x.contiguous()
        T, I = x.shape
        d = I // heads
        g = gamma.reshape(heads, d).contiguous()
        y = torch.empty_like(x)
        r = torch.empty(T * heads, dtype=torch.float32, device=x.device)
        K.rmsnorm_heads_fwd(x, I, g, y, I, r, T, heads, d)
        ctx.save_for_backward(x, g, r)
        ctx.meta = (heads, d, gamma.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g, r = ctx.saved_tensors
        heads, d, gshape = ctx.meta
        T, I = x.shape
        dy = Fn._to(dy, x.dtype)
        dx = torch.empty_like(x)
        dg = torch.empty_like(g)
        part = torch.empty(K.rmsnorm_heads_partials(T, heads, d), dtype=torch.float32, device=x.device)
        K.rmsnorm_heads_bwd(dy, I, x, I, g, r, dx, I, dg, part, T, heads, d)
        return dx, dg.view(gshape), None


class _SplitQKVFn(torch.autograd.Function):
    """merged (M, 3I) projection -> q (M, I) and k | v (M, 2I) as contiguous matrices (column-block copies)."""

    @staticmethod
    def forward(ctx, qkv, inner: int):
        qkv = qkv.contiguous()
        M = qkv.shape[0]
        q = torch.empty((M, inner), dtype=qkv.dtype, device=qkv.device)
        kv = torch.empty((M, 2 * inner), dtype=qkv.dtype, device=qkv.device)
        K.copy_cols(qkv, 3 * inner, q, inner, M, inner, inner)
        K.copy_cols(qkv[:, inner:], 3 * inner, kv, 2 * inner, M, 2 * inner, 2 * inner)
        ctx.inner = inner
        return q, kv

    @staticmethod
    def backward(ctx, dq, dkv):
        I = ctx.inner
        dq = dq.contiguous(); dkv = dkv.contiguous()
        M = dq.shape[0]
        d = torch.empty((M, 3 * I), dtype=dq.dtype, device=dq.device)
        K.copy_cols(dq, I, d, 3 * I, M, I, I)
        K.copy_cols(dkv, 2 * I, d[:, I:], 3 * I, M, 2 * I, 2 * I)
        return d, None


class Attention(nn.Module):
    def __init__(self, dim, heads=8, dim_head=64):
        super().__init__()
        inner = dim_head * heads
        self.heads = heads
        self.norm = Fn.LayerNorm(dim)
        self.attend = Fn.Softmax(dim=-1)
        self.q_norm = RMSNorm(heads, dim_head)
        self.k_norm = RMSNorm(heads, dim_head)
        self.to_qkv = Fn.Linear(dim, inner * 3, bias=False)
        self.to_out = Fn.Linear(inner, dim, bias=False)

    def forward(self, x):
        B, N, D = x.shape
        inner = self.to_qkv.weight.shape[0] // 3
        qkv = self.to_qkv(self.norm(x))
        q, kv = _SplitQKVFn.apply(qkv.reshape(B * N, 3 * inner), inner)
        out = _QKNormAttnFn.apply(q, kv, self.q_norm.gamma, self.k_norm.gamma, uniform_segments(B, N, x.device), self.heads)
        return self.to_out(out.view(B, N, inner))


class Transformer(nn.Module):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim):
        super().__init__()
        self.norm = Fn.LayerNorm(dim)
        self.layers = nn.ModuleList([])
        for _ in range(depth):
            self.layers.append(nn.ModuleList([Attention(dim, heads=heads, dim_head=dim_head), FeedForward(dim, mlp_dim)]))

    def forward(self, x):
        for attn, ff in self.layers:
            x = Fn.AddFn.apply(attn(x), x)
            x = Fn.AddFn.apply(ff(x), x)
        return self.norm(x)


class SimpleViT(nn.Module):
    def __init__(self, *, image_size, patch_size, num_classes, dim, depth, heads, mlp_dim, channels=3, dim_head=64):
        super().__init__()
        (ih, iw), (ph, pw) = pair(image_size), pair(patch_size)
        assert ih % ph == 0 and iw % pw == 0, 'Image dimensions must be divisible by the patch size.'
        self.patch_size = (ph, pw)
        patch_dim = channels * ph * pw
        self.to_patch_embedding = nn.Sequential(Fn.Patchify(ph, pw), Fn.LayerNorm(patch_dim), Fn.Linear(patch_dim, dim), Fn.LayerNorm(dim))
        self.pos_embedding = posemb_sincos_2d(h=ih // ph, w=iw // pw, dim=dim)
        self.transformer = Transformer(dim, depth, heads, dim_head, mlp_dim)
        self.pool = "mean"
        self.to_latent = nn.Identity()
        self.linear_head = Fn.LayerNorm(dim)        # sic: simple_vit_with_qk_norm.py:128

    @Fn.autocast_aware
    def forward(self, img):
        x = self.to_patch_embedding(img)
        x = Fn.ConcatTokensFn.apply(x, None, self.pos_embedding.to(x.device, dtype=x.dtype), False)      # x += pos (simple_vit_with_qk_norm.py:134)
        x = self.transformer(x)
        x = Fn.MeanTokensFn.apply(x)
        x = self.to_latent(x)
        return self.linear_head(x)


Fn.eager_modules(globals())
