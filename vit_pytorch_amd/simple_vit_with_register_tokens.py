"""SimpleViT with register tokens on libvitk kernels -- the module contract of vit_pytorch/simple_vit_with_register_tokens.py.

`register_tokens` (num_register_tokens, dim) are learned tokens appended to every image's patch sequence before the transformer and
dropped again before the mean pool (simple_vit_with_register_tokens.py:102, 113-121).  Attention / FeedForward / Transformer are
those of SimpleViT, so the stack runs in the fused engine (engine.TransformerFn) with N = patches + registers.

The registers FOLLOW the patch tokens, as in the reference's `pack([x, r], 'b * d')` (since round 6: rounds 4-5 placed them in front --
equivalent for the outputs, but a forward hook on the transformer saw another token order than under the reference); one kernel
(`vitk_concat_tokens`, F < 0) builds the sequence and adds the positional table.
"""
from __future__ import annotations

import torch
from torch import nn

from . import functional as Fn
from .simple_vit import Attention, FeedForward, Transformer, posemb_sincos_2d  # noqa: F401
from .vit import pair


class SimpleViT(nn.Module):
    def __init__(self, *, image_size, patch_size, num_classes, dim, depth, heads, mlp_dim, num_register_tokens=4, channels=3, dim_head=64):
        super().__init__()
        (ih, iw), (ph, pw) = pair(image_size), pair(patch_size)
        assert ih % ph == 0 and iw % pw == 0, 'Image dimensions must be divisible by the patch size.'
        self.patch_size = (ph, pw)
        patch_dim = channels * ph * pw
        self.to_patch_embedding = nn.Sequential(Fn.Patchify(ph, pw), Fn.LayerNorm(patch_dim), Fn.Linear(patch_dim, dim), Fn.LayerNorm(dim))
        self.register_tokens = nn.Parameter(torch.randn(num_register_tokens, dim))
        self.pos_embedding = posemb_sincos_2d(h=ih // ph, w=iw // pw, dim=dim)
        self.transformer = Transformer(dim, depth, heads, dim_head, mlp_dim)
        self.pool = "mean"
        self.to_latent = nn.Identity()
        self.linear_head = Fn.Linear(dim, num_classes)

    def _pos_on(self, device, dtype):
        """The sincos table with R zero rows behind it (it is added to the patch tokens only), built ONCE per (device, dtype) -- the
        reference converts / concatenates on every forward (simple_vit_with_register_tokens.py:139-142); no torch op on the data path."""
        slot = self.__dict__.setdefault("_pos_cache", {})
        key = (str(device), dtype)
        if key not in slot:
            slot.clear()
            pos = self.pos_embedding.to(device, dtype=dtype)
            R = self.register_tokens.shape[0]
            slot[key] = torch.cat([pos, pos.new_zeros(R, pos.shape[1])], dim=0).contiguous()
        return slot[key]

    @Fn.autocast_aware
    def forward(self, img):
        x = self.to_patch_embedding(img)
        R = self.register_tokens.shape[0]
        pos = self._pos_on(x.device, x.dtype)
        P = x.shape[1]
        x = Fn.ConcatTokensFn.apply(x, self.register_tokens, pos, True)         # (B, patches + R, dim): pack([x, r])
        x = self.transformer(x)
        x = Fn.TokenSliceFn.apply(x, 0, P) if R else x                          # unpack: the patch tokens
        x = Fn.MeanTokensFn.apply(x)
        x = self.to_latent(x)
        return self.linear_head(x)


Fn.eager_modules(globals())
