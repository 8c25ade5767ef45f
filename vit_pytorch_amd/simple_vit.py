"""Drop-in for vit_pytorch/simple_vit.py (posemb_sincos_2d :12-21, FeedForward :25-35,
Attention :37-62, Transformer :64-78, SimpleViT :80-120) on libvitk kernels."""
from __future__ import annotations

import torch
from torch import nn

from . import engine as E
from . import functional as Fn
from .vit import _has_fwd_hooks, pair


def posemb_sincos_2d(h, w, dim, temperature: int = 10000, dtype=torch.float32):
    """Host-side constant table (simple_vit.py:12-21): [sin x, cos x, sin y, cos y], x = column index."""
    y, x = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    assert (dim % 4) == 0, "feature dimension must be multiple of 4 for sincos emb"
    omega = torch.arange(dim // 4) / (dim // 4 - 1)
    omega = 1.0 / (temperature ** omega)

    y = y.flatten()[:, None] * omega[None, :]
    x = x.flatten()[:, None] * omega[None, :]
    pe = torch.cat((x.sin(), x.cos(), y.sin(), y.cos()), dim=1)
    return pe.type(dtype)


class FeedForward(nn.Module):
    def __init__(self, dim, hidden_dim):
        super().__init__()
        self.net = nn.Sequential(
            Fn.LayerNorm(dim),
            Fn.Linear(dim, hidden_dim),
            Fn.GELU(),
            Fn.Linear(hidden_dim, dim),
        )

    def forward(self, x):
        return self.net(x)


class Attention(nn.Module):
    def __init__(self, dim, heads=8, dim_head=64):
        super().__init__()
        inner_dim = dim_head * heads
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.norm = Fn.LayerNorm(dim)

        self.attend = Fn.Softmax(dim=-1)

        self.to_qkv = Fn.Linear(dim, inner_dim * 3, bias=False)
        self.to_out = Fn.Linear(inner_dim, dim, bias=False)

    def _needs_attention_matrix(self) -> bool:
        return _has_fwd_hooks(self.attend)

    def forward(self, x):
        x = self.norm(x)
        qkv = self.to_qkv(x)
        if self._needs_attention_matrix():
            dots = Fn.ScoresFn.apply(qkv, self.heads, self.scale)
            attn = self.attend(dots)
            out = Fn.AttnValuesFn.apply(attn, qkv, self.heads)
        else:
            out = Fn.FusedAttnFn.apply(qkv, self.heads, self.scale)
        return self.to_out(out)


class Transformer(nn.Module):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim):
        super().__init__()
        self.norm = Fn.LayerNorm(dim)
        self.layers = nn.ModuleList([])
        self._heads, self._dim_head = heads, dim_head
        for _ in range(depth):
            self.layers.append(nn.ModuleList([
                Attention(dim, heads=heads, dim_head=dim_head),
                FeedForward(dim, mlp_dim),
            ]))

    def _fusable(self) -> bool:
        for attn, ff in self.layers:
            if any(_has_fwd_hooks(m) for m in list(attn.modules()) + list(ff.modules())):
                return False
        return not _has_fwd_hooks(self.norm)

    def forward(self, x):
        if self._fusable():
            params = []
            for attn, ff in self.layers:
                params += E.pack_layer_params(attn, ff)
            return E.TransformerFn.apply(x, self._heads, self._dim_head, 0.0, 0, getattr(self, "_fp8", None), self.norm.weight, self.norm.bias, *params)
        x = Fn._to(x, self.norm.weight.dtype)
        for attn, ff in self.layers:
            x = Fn.AddFn.apply(attn(x), x)
            x = Fn.AddFn.apply(ff(x), x)
        return self.norm(x)


class SimpleViT(nn.Module):
    def __init__(self, *, image_size, patch_size, num_classes, dim, depth, heads, mlp_dim, channels=3, dim_head=64):
        super().__init__()
        image_height, image_width = pair(image_size)
        self.patch_size = patch_height, patch_width = pair(patch_size)

        assert image_height % patch_height == 0 and image_width % patch_width == 0, 'Image dimensions must be divisible by the patch size.'

        patch_dim = channels * patch_height * patch_width

        self.to_patch_embedding = nn.Sequential(
            Fn.Patchify(patch_height, patch_width),
            Fn.LayerNorm(patch_dim),
            Fn.Linear(patch_dim, dim),
            Fn.LayerNorm(dim),
        )

        # plain tensor attribute, as in the reference: not a parameter, not a buffer, not in state_dict
        self.pos_embedding = posemb_sincos_2d(
            h=image_height // patch_height,
            w=image_width // patch_width,
            dim=dim,
        )

        self.transformer = Transformer(dim, depth, heads, dim_head, mlp_dim)

        self.pool = "mean"
        self.to_latent = nn.Identity()

        self.linear_head = Fn.Linear(dim, num_classes)

    def _pos_on(self, device, dtype):
        # simple_vit.py:114: `self.pos_embedding.to(device, dtype=x.dtype)` every forward; cached here per (device, dtype)
        key = (str(device), dtype)
        cache = self.__dict__.setdefault("_pos_cache", {})
        if key not in cache:
            cache.clear()
            cache[key] = self.pos_embedding.to(device, dtype=dtype).contiguous()
        return cache[key]

    def forward(self, img):
        pe = self.to_patch_embedding
        w = pe[2].weight
        pos = self._pos_on(img.device, w.dtype)
        if not any(_has_fwd_hooks(m) for m in pe.modules()):
            x = E.PatchEmbedFn.apply(img, pe[0].p1, pe[0].p2, pe[1].weight, pe[1].bias, pe[2].weight, pe[2].bias,
                                     pe[3].weight, pe[3].bias, None, pos)
        else:
            x = pe(img)
            x = Fn.AddFn.apply(x, pos.unsqueeze(0).expand_as(x).contiguous())

        x = self.transformer(x)

        if not (_has_fwd_hooks(self.to_latent) or _has_fwd_hooks(self.linear_head)):
            return E.HeadFn.apply(x, True, self.linear_head.weight, self.linear_head.bias)
        x = Fn.MeanTokensFn.apply(x)
        x = self.to_latent(x)
        return self.linear_head(x)
