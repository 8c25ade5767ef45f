"""ViT with patch dropout on libvitk kernels -- the module contract of vit_pytorch/vit_with_patch_dropout.py.

This file's ViT is the older layout of the reference: the patch embedding is Rearrange + Linear (no LayerNorms), `pos_embedding`
has one row per patch (no cls row) and is added BEFORE the cls token is concatenated, the transformer has no final LayerNorm and
`mlp_head` is LayerNorm + Linear (vit_with_patch_dropout.py:59-78, 80-99).  `PatchDropout` keeps a random subset of
max(1, int(n (1 - prob))) patch tokens per image in training mode (:14-32): the indices come from torch's RNG exactly as in the
reference (topk of a normal draw), the gather and its adjoint scatter are `vitk_gather_tokens`.
"""
from __future__ import annotations

import torch
from torch import nn

from . import functional as Fn
from .vit import Attention, FeedForward, pair  # noqa: F401  (same blocks as vit.py: pre-norm attention, LayerNorm-first FeedForward)


class PatchDropout(nn.Module):
    def __init__(self, prob):
        super().__init__()
        assert 0 <= prob < 1.
        self.prob = prob

    def forward(self, x):
        if not self.training or self.prob == 0.:
            return x
        b, n, _ = x.shape
        keep_n = max(1, int(n * (1 - self.prob)))
        keep = torch.randn(b, n, device=x.device).topk(keep_n, dim=-1).indices       # index selection only; no token data touched
        return Fn.GatherTokensFn.apply(x, keep)


class Transformer(nn.Module):
    """attn(x) + x ; ff(x) + x per layer, NO final LayerNorm (vit_with_patch_dropout.py:82-95)."""

    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout=0.):
        super().__init__()
        self.layers = nn.ModuleList([])
        for _ in range(depth):
            self.layers.append(nn.ModuleList([Attention(dim, heads=heads, dim_head=dim_head, dropout=dropout),
                                              FeedForward(dim, mlp_dim, dropout=dropout)]))

    def forward(self, x):
        for attn, ff in self.layers:
            x = Fn.AddFn.apply(attn(x), x)
            x = Fn.AddFn.apply(ff(x), x)
        return x


class ViT(nn.Module):
    def __init__(self, *, image_size, patch_size, num_classes, dim, depth, heads, mlp_dim, pool='cls', channels=3, dim_head=64,
                 dropout=0., emb_dropout=0., patch_dropout=0.25):
        super().__init__()
        (ih, iw), (ph, pw) = pair(image_size), pair(patch_size)
        assert ih % ph == 0 and iw % pw == 0, 'Image dimensions must be divisible by the patch size.'
        assert pool in {'cls', 'mean'}, 'pool type must be either cls (cls token) or mean (mean pooling)'
        self.patch_size = (ph, pw)
        num_patches = (ih // ph) * (iw // pw)
        patch_dim = channels * ph * pw
        self.to_patch_embedding = nn.Sequential(Fn.Patchify(ph, pw), Fn.Linear(patch_dim, dim))
        self.pos_embedding = nn.Parameter(torch.randn(num_patches, dim))
        self.cls_token = nn.Parameter(torch.randn(1, 1, dim))
        self.patch_dropout = PatchDropout(patch_dropout)
        self.dropout = Fn.Dropout(emb_dropout)
        self.transformer = Transformer(dim, depth, heads, dim_head, mlp_dim, dropout)
        self.pool = pool
        self.to_latent = nn.Identity()
        self.mlp_head = nn.Sequential(Fn.LayerNorm(dim), Fn.Linear(dim, num_classes))

    @Fn.autocast_aware
    def forward(self, img):
        x = self.to_patch_embedding(img)
        x = Fn.ConcatTokensFn.apply(x, None, self.pos_embedding, False)                 # x += pos_embedding
        x = self.patch_dropout(x)
        x = Fn.ConcatTokensFn.apply(x, self.cls_token.view(1, -1), None, False)         # cat(cls, x)
        x = self.dropout(x)
        x = self.transformer(x)
        x = Fn.MeanTokensFn.apply(x) if self.pool == 'mean' else _first_token(x)
        x = self.to_latent(x)
        return self.mlp_head(x)


def _first_token(x):
    from .vit import _ClsRowFn
    return _ClsRowFn.apply(x)


Fn.eager_modules(globals())
