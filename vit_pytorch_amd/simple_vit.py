"""SimpleViT on libvitk kernels -- the module contract of vit_pytorch/simple_vit.py.

What a user of the reference relies on and what is therefore reproduced exactly:
constructor keywords (simple_vit.py:81), `state_dict` keys / shapes / order, the module tree
(`to_patch_embedding[0..3]`, `transformer.layers[i][0|1]` with `norm / attend / to_qkv / to_out` and `net[0..3]`,
`to_latent`, `linear_head`), `pool == "mean"`, and `pos_embedding` being a plain tensor attribute (not in `state_dict`).
The arithmetic behind those names runs in the fused engine (engine.py); sub-modules remain callable one by one and
forward hooks on any of them switch the enclosing block to the op-by-op path.
"""
from __future__ import annotations

import torch
from torch import nn

from ._epoch import note_grad_mode
from . import engine as E
from . import functional as Fn
from .vit import _has_fwd_hooks, pair


def posemb_sincos_2d(h, w, dim, temperature: int = 10000, dtype=torch.float32):
    """The fixed 2-D sin/cos table (simple_vit.py:12-21): row = patch (row-major grid), columns =
    [sin(x w_k), cos(x w_k), sin(y w_k), cos(y w_k)] with x the grid column, w_k = temperature^(-k / (dim/4 - 1))."""
    assert (dim % 4) == 0, "feature dimension must be multiple of 4 for sincos emb"
    quarter = dim // 4
    omega = 1.0 / (temperature ** (torch.arange(quarter) / (quarter - 1)))
    rows, cols = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    ang_x = cols.reshape(-1, 1) * omega.reshape(1, -1)
    ang_y = rows.reshape(-1, 1) * omega.reshape(1, -1)
    table = torch.cat([ang_x.sin(), ang_x.cos(), ang_y.sin(), ang_y.cos()], dim=1)
    return table.type(dtype)


def _any_hooks(*modules) -> bool:
    return any(_has_fwd_hooks(sub) for m in modules for sub in m.modules())


class FeedForward(nn.Module):
    """LayerNorm -> Linear -> GELU -> Linear (no dropout in SimpleViT); `net` indices 0, 1, 3 carry parameters."""

    def __init__(self, dim, hidden_dim):
        super().__init__()
        stages = [Fn.LayerNorm(dim), Fn.Linear(dim, hidden_dim), Fn.GELU(), Fn.Linear(hidden_dim, dim)]
        self.net = nn.Sequential(*stages)

    def forward(self, x):
        return self.net(x)


class Attention(nn.Module):
    """Pre-norm multi-head attention with a bias-free output projection (simple_vit.py:37-62)."""

    def __init__(self, dim, heads=8, dim_head=64):
        super().__init__()
        width = heads * dim_head
        self.heads, self.scale = heads, dim_head ** -0.5
        self.norm = Fn.LayerNorm(dim)
        self.attend = Fn.Softmax(dim=-1)
        self.to_qkv = Fn.Linear(dim, 3 * width, bias=False)
        self.to_out = Fn.Linear(width, dim, bias=False)

    def forward(self, x):
        qkv = self.to_qkv(self.norm(x))                 # merged q | k | v, consumed in place by the attention kernels
        if _has_fwd_hooks(self.attend):                 # someone wants the N x N matrix (Recorder-style hooks)
            probs = self.attend(Fn.ScoresFn.apply(qkv, self.heads, self.scale))
            mixed = Fn.AttnValuesFn.apply(probs, qkv, self.heads)
        else:
            mixed = Fn.FusedAttnFn.apply(qkv, self.heads, self.scale)
        return self.to_out(mixed)


class Transformer(nn.Module):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim):
        super().__init__()
        self._heads, self._dim_head = heads, dim_head
        self.norm = Fn.LayerNorm(dim)
        self.layers = nn.ModuleList(
            nn.ModuleList([Attention(dim, heads=heads, dim_head=dim_head), FeedForward(dim, mlp_dim)]) for _ in range(depth))

    def _fusable(self) -> bool:
        """Nothing observes the inside of the stack, and every block is exactly what the constructor builds (simple_vit.py:23-62) -- model
        surgery (a LoRA / quantised Linear subclass, an extra module, a foreign block) runs op by op through the modules that are there."""
        for pair_ in self.layers:
            if len(pair_) != 2:
                return False
            attn, ff = pair_
            if type(attn) is not Attention or type(ff) is not FeedForward:
                return False
            net = ff.net
            if not (Fn.plain_layernorm(attn.norm) and Fn.plain_linear(attn.to_qkv) and attn.to_qkv.bias is None and Fn.exactly(attn.attend, Fn.Softmax)
                    and Fn.plain_linear(attn.to_out) and attn.to_out.bias is None and type(net) is nn.Sequential and len(net) == 4
                    and Fn.plain_layernorm(net[0]) and Fn.plain_linear(net[1]) and Fn.exactly(net[2], Fn.GELU) and Fn.plain_linear(net[3])):
                return False
        if not Fn.plain_layernorm(self.norm):
            return False
        return not _any_hooks(self.norm, *(blk for pair_ in self.layers for blk in pair_))

    def wants_16bit_stream(self, B: int, N: int) -> bool:
        """See vit.Transformer.wants_16bit_stream: the patch embedding then writes the stream in the parameter dtype directly."""
        if not len(self.layers) or not self._fusable():
            return False
        attn, ff = self.layers[0]
        lp = E.pack_layer_params(attn, ff)
        return E.forward_stream_is_16bit(self.norm.weight.dtype, B * N, self.norm.weight.shape[0], self._heads * self._dim_head, lp[7].shape[0],
                                         len(self.layers), 0.0, getattr(self, "_fp8", None), lp[3] is not None, lp[8] is not None)

    @Fn.autocast_aware
    def forward(self, x):
        if not self._fusable():
            x = Fn.cast(x, self.norm.weight.dtype)       # in the graph: the embedding stage may have produced an f32 stream
            for attn, ff in self.layers:
                x = Fn.AddFn.apply(attn(x), x)
                x = Fn.AddFn.apply(ff(x), x)
            return self.norm(x)
        flat = [t for attn, ff in self.layers for t in E.pack_layer_params(attn, ff)]
        note_grad_mode(torch.is_grad_enabled())      # Function.forward cannot see no_grad(): it decides what to keep from this
        return E.TransformerFn.apply(x, self._heads, self._dim_head, 0.0, 0, getattr(self, "_fp8", None),
                                     self.norm.weight, self.norm.bias, *flat)


class SimpleViT(nn.Module):
    def __init__(self, *, image_size, patch_size, num_classes, dim, depth, heads, mlp_dim, channels=3, dim_head=64):
        super().__init__()
        (ih, iw), (ph, pw) = pair(image_size), pair(patch_size)
        assert ih % ph == 0 and iw % pw == 0, 'Image dimensions must be divisible by the patch size.'
        self.patch_size = (ph, pw)
        patch_dim = channels * ph * pw
        self.to_patch_embedding = nn.Sequential(
            Fn.Patchify(ph, pw), Fn.LayerNorm(patch_dim), Fn.Linear(patch_dim, dim), Fn.LayerNorm(dim))
        # a plain tensor attribute as in the reference: no parameter, no buffer, absent from state_dict
        self.pos_embedding = posemb_sincos_2d(h=ih // ph, w=iw // pw, dim=dim)
        self.transformer = Transformer(dim, depth, heads, dim_head, mlp_dim)
        self.pool = "mean"
        self.to_latent = nn.Identity()
        self.linear_head = Fn.Linear(dim, num_classes)

    def _pos_on(self, device, dtype):
        # simple_vit.py:114 converts the table on every forward; here once per (device, dtype)
        slot = self.__dict__.setdefault("_pos_cache", {})
        key = (str(device), dtype)
        if key not in slot:
            slot.clear()
            slot[key] = self.pos_embedding.to(device, dtype=dtype).contiguous()
        return slot[key]

    @Fn.autocast_aware
    def forward(self, img):
        embed = self.to_patch_embedding
        # exactly Rearrange -> LayerNorm -> Linear -> LayerNorm as built (simple_vit.py:90-95); a stem the user swapped in is simply called
        built = (type(embed) is nn.Sequential and len(embed) == 4 and Fn.exactly(embed[0], Fn.Patchify) and Fn.plain_layernorm(embed[1])
                 and Fn.plain_linear(embed[2]) and Fn.plain_layernorm(embed[3]))
        pos = self._pos_on(img.device, embed[2].weight.dtype if built else next(self.transformer.parameters()).dtype)
        if not built or _any_hooks(embed) or embed[1].weight.shape[0] % 4:     # surgery, hooks, or a patch_dim off the fused stage's 16-byte rows (147): op by op
            tokens = embed(img)
            tokens = Fn.AddFn.apply(tokens, pos.unsqueeze(0).expand_as(tokens).contiguous())
        else:
            ntok = (img.shape[-2] // embed[0].p1) * (img.shape[-1] // embed[0].p2)
            tokens = E.PatchEmbedFn.apply(img, embed[0].p1, embed[0].p2, embed[1].weight, embed[1].bias, embed[2].weight,
                                          embed[2].bias, embed[3].weight, embed[3].bias, None, pos,
                                          img.dim() == 4 and getattr(self.transformer, "wants_16bit_stream", lambda b, n: False)(img.shape[0], ntok))
        tokens = self.transformer(tokens)
        if (_has_fwd_hooks(self.to_latent) or _has_fwd_hooks(self.linear_head) or not Fn.plain_linear(self.linear_head)
                or type(self.to_latent) is not nn.Identity):        # hooks, or a head / to_latent the user swapped in: call them (simple_vit.py:117-120)
            return self.linear_head(self.to_latent(Fn.MeanTokensFn.apply(tokens)))
        return E.HeadFn.apply(tokens, True, self.linear_head.weight, self.linear_head.bias)


Fn.eager_modules(globals())
