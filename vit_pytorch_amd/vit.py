"""Drop-in for vit_pytorch/vit.py (FeedForward :15-28, Attention :30-64, Transformer :66-83,
ViT :85-138): same constructors, module tree, children order, parameter names and shapes,
same forward() contract -- executed by hand-written gfx950 kernels (libvitk) on an MI355X.

There is no CPU path: calling forward() on CPU tensors raises (vit_pytorch_amd._lib.VitkError).
"""
from __future__ import annotations

import torch
from torch import nn
from torch.nn import Module, ModuleList

from ._epoch import note_grad_mode
from . import engine as E
from . import functional as Fn


def pair(t):
    return t if isinstance(t, tuple) else (t, t)


def _has_fwd_hooks(m: Module) -> bool:
    return Fn.observed(m)


def _pristine_block(attn, ff) -> bool:
    """The (attention, feed-forward) pair is exactly what the constructor builds (vit.py:15-64): only then may the fused stage read the
    parameters and skip the modules.  Model surgery -- a LoRA / quantised Linear subclass in place of to_qkv, an extra module in `net`, a
    foreign block appended to `layers` -- takes the op-by-op path, which calls whatever modules are there."""
    if type(attn) is not Attention or type(ff) is not FeedForward:
        return False
    if not (Fn.plain_layernorm(attn.norm) and Fn.plain_linear(attn.to_qkv) and attn.to_qkv.bias is None
            and Fn.exactly(attn.attend, Fn.Softmax) and Fn.exactly(attn.dropout, Fn.Dropout)):
        return False
    out = attn.to_out
    if not (type(out) is nn.Identity or (type(out) is nn.Sequential and len(out) == 2 and Fn.plain_linear(out[0]) and Fn.exactly(out[1], Fn.Dropout))):
        return False
    net = ff.net
    return (type(net) is nn.Sequential and len(net) == 6 and Fn.plain_layernorm(net[0]) and Fn.plain_linear(net[1]) and Fn.exactly(net[2], Fn.GELU)
            and Fn.exactly(net[3], Fn.Dropout) and Fn.plain_linear(net[4]) and Fn.exactly(net[5], Fn.Dropout))


class FeedForward(Module):
    """LayerNorm -> Linear -> GELU -> Dropout -> Linear -> Dropout (vit.py:15-28); `net` indices 0, 1, 4 carry parameters."""

    def __init__(self, dim, hidden_dim, dropout=0.):
        super().__init__()
        stages = [Fn.LayerNorm(dim), Fn.Linear(dim, hidden_dim), Fn.GELU(), Fn.Dropout(dropout),
                  Fn.Linear(hidden_dim, dim), Fn.Dropout(dropout)]
        self.net = nn.Sequential(*stages)

    def forward(self, x):
        return self.net(x)


class Attention(Module):
    """Pre-norm multi-head attention (vit.py:30-64).  `to_out` is Linear + Dropout, or Identity when a single head already
    spans the model width; `attend` / `dropout` exist as modules so that hooks and `.p` edits keep working."""

    def __init__(self, dim, heads=8, dim_head=64, dropout=0.):
        super().__init__()
        width = heads * dim_head
        self.heads, self.scale = heads, dim_head ** -0.5
        self.norm = Fn.LayerNorm(dim)
        self.attend = Fn.Softmax(dim=-1)
        self.dropout = Fn.Dropout(dropout)
        self.to_qkv = Fn.Linear(dim, 3 * width, bias=False)
        single_full_head = heads == 1 and dim_head == dim
        self.to_out = nn.Identity() if single_full_head else nn.Sequential(Fn.Linear(width, dim), Fn.Dropout(dropout))

    def _needs_attention_matrix(self) -> bool:
        # forward hooks on `attend` (recorder.py:26-29) or active attention dropout need the N x N matrix
        return _has_fwd_hooks(self.attend) or (self.training and self.dropout.p > 0.)

    def forward(self, x):
        x = self.norm(x)
        qkv = self.to_qkv(x)  # (b, n, 3*h*d): q | k | v, each '(h d)' with h outer -- consumed in place
        if self._needs_attention_matrix():
            dots = Fn.ScoresFn.apply(qkv, self.heads, self.scale)   # matmul(q, k^T) * scale  (vit.py:57)
            attn = self.attend(dots)
            attn = self.dropout(attn)
            out = Fn.AttnValuesFn.apply(attn, qkv, self.heads)      # matmul(attn, v) + merge heads (vit.py:62-63)
        else:
            out = Fn.FusedAttnFn.apply(qkv, self.heads, self.scale)
        return self.to_out(out)


class Transformer(Module):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout=0.):
        super().__init__()
        self.norm = Fn.LayerNorm(dim)
        self.layers = ModuleList([])
        self._heads, self._dim_head = heads, dim_head
        self._drop_calls, self._drop_salt = 0, Transformer._instances[0]
        Transformer._instances[0] += 1

        for _ in range(depth):
            self.layers.append(ModuleList([
                Attention(dim, heads=heads, dim_head=dim_head, dropout=dropout),
                FeedForward(dim, mlp_dim, dropout=dropout),
            ]))

    # every training call draws fresh dropout seeds from torch.initial_seed(), THIS module's call count and its construction
    # index (instance state: two models in one process neither interleave their sequences nor share masks); not in state_dict
    _instances = [0]

    def _drop_state(self):
        """(calls so far, per-instance salt) of the fused-dropout seed sequence.  Plain instance state: modules unpickled from an older
        torch.save(model) lack it (defaults here), and copy.deepcopy re-salts (__deepcopy__) so that an EMA / teacher copy does not
        draw the masks of its source."""
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
            ps.update(m.p for m in list(ff.net) + list(attn.modules()) if isinstance(m, nn.Dropout))
        return ps.pop() if len(ps) == 1 else (0.0 if not ps else None)

    def _fusable(self, x) -> bool:
        """The fused engine covers the block when nothing needs to observe its inside; active dropout is fused too when
        the shapes are those the fused-dropout kernels serve (engine.dropout_fusable), else the block runs op by op."""
        for pair_ in self.layers:
            if len(pair_) != 2 or not _pristine_block(pair_[0], pair_[1]):
                return False
            attn, ff = pair_
            if any(_has_fwd_hooks(m) for m in list(attn.modules()) + list(ff.modules())):
                return False
        if _has_fwd_hooks(self.norm) or not Fn.plain_layernorm(self.norm):
            return False
        # widths that are not multiples of 4 (T2T-ViT's token-to-token layers: 147, 1323 -- t2t.py:45) run op by op on the
        # any-width kernels; the fused engine's vector kernels need 16-byte rows
        widths = [self.norm.weight.shape[0]]
        for attn, ff in self.layers:
            widths += [attn.to_qkv.weight.shape[0], ff.net[1].weight.shape[0]]
        if any(w % 4 for w in widths):
            return False
        p = self._dropout_p()
        if p is None:
            return False
        if p > 0.:
            if not len(self.layers) or isinstance(self.layers[0][0].to_out, nn.Identity):
                return False
            B, N, D = x.shape
            return E.dropout_fusable(self.norm.weight.dtype, B, N, D, self._heads, self._dim_head, self.layers[0][1].net[1].weight.shape[0])
        return True

    def wants_16bit_stream(self, B: int, N: int) -> bool:
        """Whether forward() on a (B, N, dim) input will run the fused stage with its residual stream in the parameter dtype: the patch
        embedding in front of it then writes that dtype directly (engine.PatchEmbedFn out16) instead of float32 + a cast."""
        if not len(self.layers):
            return False
        import types
        D = self.norm.weight.shape[0]
        if not self._fusable(types.SimpleNamespace(shape=(B, N, D))):
            return False
        attn, ff = self.layers[0]
        return E.forward_stream_is_16bit(self.norm.weight.dtype, B * N, D, self._heads * self._dim_head, ff.net[1].weight.shape[0], len(self.layers),
                                         float(self._dropout_p()), getattr(self, "_fp8", None), not isinstance(attn.to_out, nn.Identity),
                                         ff.net[1].bias is not None)

    @Fn.autocast_aware           # the block on its own inside an autocast region (t2t.py:45,57 builds its layers from it)
    def forward(self, x):
        if self._fusable(x):
            params = []
            for attn, ff in self.layers:
                params += E.pack_layer_params(attn, ff)
            p = self._dropout_p()
            seed = 0
            if p > 0.:
                seed = (int(torch.initial_seed()) + 0x9E3779B1 * self._drop_state()[0] + 0x85EBCA6B * self._drop_state()[1] + 0x632BE5AB * Fn.dist_rank()) & 0xffffffff
                self.__dict__["_drop_calls"] = self._drop_state()[0] + 1
            note_grad_mode(torch.is_grad_enabled())      # Function.forward cannot see no_grad(): it decides what to keep from this
            return E.TransformerFn.apply(x, self._heads, self._dim_head, float(p), seed, getattr(self, "_fp8", None), self.norm.weight, self.norm.bias, *params)
        x = Fn.cast(x, self.norm.weight.dtype)       # in the graph: the embedding stage may have produced an f32 stream
        for attn, ff in self.layers:
            x = Fn.AddFn.apply(attn(x), x)
            x = Fn.AddFn.apply(ff(x), x)
        return self.norm(x)


class ViT(Module):
    def __init__(self, *, image_size, patch_size, num_classes, dim, depth, heads, mlp_dim, pool='cls', channels=3,
                 dim_head=64, dropout=0., emb_dropout=0.):
        super().__init__()
        (ih, iw), (ph, pw) = pair(image_size), pair(patch_size)
        assert ih % ph == 0 and iw % pw == 0, 'Image dimensions must be divisible by the patch size.'
        assert pool in {'cls', 'mean'}, 'pool type must be either cls (cls token) or mean (mean pooling)'
        self.patch_size = (ph, pw)
        self.pool = pool
        n_cls = int(pool == 'cls')                      # a (0, dim) cls_token parameter exists with mean pooling (vit.py:97)
        n_patches = (ih // ph) * (iw // pw)
        patch_dim = channels * ph * pw

        # children in the reference's order: to_patch_embedding, dropout, transformer, to_latent, mlp_head
        self.to_patch_embedding = nn.Sequential(
            Fn.Patchify(ph, pw), Fn.LayerNorm(patch_dim), Fn.Linear(patch_dim, dim), Fn.LayerNorm(dim))
        self.cls_token = nn.Parameter(torch.randn(n_cls, dim))
        self.pos_embedding = nn.Parameter(torch.randn(n_patches + n_cls, dim))
        self.dropout = Fn.Dropout(emb_dropout)
        self.transformer = Transformer(dim, depth, heads, dim_head, mlp_dim, dropout)
        self.to_latent = nn.Identity()
        self.mlp_head = Fn.Linear(dim, num_classes) if num_classes > 0 else None

    def _embed_fusable(self) -> bool:
        pe = self.to_patch_embedding
        # exactly Rearrange -> LayerNorm -> Linear -> LayerNorm as built (vit.py:99-104); a stem the user swapped in is simply called
        if not (type(pe) is nn.Sequential and len(pe) == 4 and Fn.exactly(pe[0], Fn.Patchify) and Fn.plain_layernorm(pe[1])
                and Fn.plain_linear(pe[2]) and Fn.plain_layernorm(pe[3]) and isinstance(self.dropout, nn.Dropout)):
            return False
        if any(_has_fwd_hooks(m) for m in pe.modules()):
            return False
        if pe[1].weight.shape[0] % 4 or pe[2].weight.shape[0] % 4:   # patch_dim (7 x 7 x 3 = 147) or dim (ViT(dim = 30)) off the 16-byte rows of the fused stage: op by op on the any-width kernels
            return False
        return not (self.training and self.dropout.p > 0.) and not _has_fwd_hooks(self.dropout)

    @Fn.autocast_aware
    def forward(self, img):
        pe = self.to_patch_embedding
        if self._embed_fusable():
            ntok = (img.shape[-2] // pe[0].p1) * (img.shape[-1] // pe[0].p2) + self.cls_token.shape[0]
            x = E.PatchEmbedFn.apply(img, pe[0].p1, pe[0].p2, pe[1].weight, pe[1].bias, pe[2].weight, pe[2].bias,
                                     pe[3].weight, pe[3].bias, self.cls_token, self.pos_embedding,
                                     img.dim() == 4 and getattr(self.transformer, "wants_16bit_stream", lambda b, n: False)(img.shape[0], ntok))
        else:
            x = pe(img)
            x = _prepend_cls_add_pos(x, self.cls_token, self.pos_embedding)
            x = self.dropout(x)

        x = self.transformer(x)

        if self.mlp_head is None:
            return x

        # the fused pool + Linear serves the modules the constructor built; a head the user swapped in (`model.mlp_head = nn.Identity()` for
        # features, a new nn.Linear / nn.Sequential for fine-tuning) or a non-trivial to_latent is simply called, as vit.py:137-138 does
        head_plain = (Fn.plain_linear(self.mlp_head) and type(self.to_latent) is nn.Identity
                      and not (_has_fwd_hooks(self.to_latent) or _has_fwd_hooks(self.mlp_head)))
        if head_plain:
            return E.HeadFn.apply(x, self.pool == 'mean', self.mlp_head.weight, self.mlp_head.bias)
        x = Fn.MeanTokensFn.apply(x) if self.pool == 'mean' else _ClsRowFn.apply(x)
        x = self.to_latent(x)
        return self.mlp_head(x)


class _ClsRowFn(torch.autograd.Function):
    """x[:, 0] (vit.py:135) as a strided device copy (no compute)."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = x.shape
        return x[:, 0].contiguous()

    @staticmethod
    def backward(ctx, g):
        dx = torch.zeros(ctx.shape, dtype=g.dtype, device=g.device)
        dx[:, 0].copy_(g)
        return dx


def _prepend_cls_add_pos(x, cls, pos):
    """cat(cls, x) + pos[:seq] (vit.py:122-127) for the non-fused embedding path: one launch for the whole batch."""
    return Fn.ConcatTokensFn.apply(x, cls, pos, False)


Fn.eager_modules(globals())
