"""Fine-grained autograd Functions and nn.Module shells (one per torch.nn op the reference uses).

These make every SUB-module of the drop-in callable on its own, exactly like the reference's
(`vit.to_patch_embedding(img)`, `vit.transformer(tokens)`, `Attention(x)`, forward hooks on
`Attention.attend` used by recorder.py:26-29, `to_latent` hooks used by dino.py:138-151), and
they are the path taken when dropout is active in training.  They run the same libvitk
kernels as the fused engine, one op at a time; the fused engine (engine.py) is the fast path.

The Module shells subclass the torch.nn classes of the reference so `state_dict()` keys, shapes
and `isinstance` checks are unchanged; only `forward` is replaced.
"""
from __future__ import annotations

import os

import torch
from torch import nn

from . import kernels as K
from . import ops
from ._lib import VitkError
from ._epoch import model_grad_scope, weight_key

Tensor = torch.Tensor
F32 = torch.float32


def _rows(x: Tensor):
    D = x.shape[-1]
    return x.numel() // D, D


def dist_rank() -> int:
    """Rank of this process (0 outside torch.distributed): data-parallel ranks draw different dropout masks."""
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def _to(x: Tensor, dtype) -> Tensor:
    """dtype conversion on the device through vitk_cast (no torch compute)."""
    x = x.contiguous()
    if x.dtype == dtype:
        return x
    y = torch.empty_like(x, dtype=dtype)
    K.cast(x, y)
    return y


class CastFn(torch.autograd.Function):
    """dtype conversion INSIDE the autograd graph (the raw `_to` is for use inside other Functions only): the gradient is
    converted back to the input's dtype.  Needed where a fused stage hands its f32 residual stream to op-by-op modules of a
    16-bit model (hooks on `attend`, Identity `to_out`, dropout at shapes the fused-dropout kernels do not serve)."""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        return _to(x, dtype)

    @staticmethod
    def backward(ctx, g):
        return _to(g, ctx.src), None


def cast(x: Tensor, dtype) -> Tensor:
    return x if x.dtype == dtype else CastFn.apply(x, dtype)


class CastParamsFn(torch.autograd.Function):
    """Every float32 parameter of a model to the 16-bit compute type in ceil(count / 64) launches (vitk_cast_many), and their 16-bit
    gradients back to float32 the same way -- ONE autograd node, so the way back is one multi-tensor conversion at the end of the
    backward.  This is what torch.autocast does to the weights of the reference's nn.Linear layers, op by op."""

    @staticmethod
    def forward(ctx, dtype, *params):
        outs = [torch.empty(p.shape, dtype=dtype, device=p.device) for p in params]
        def aligned(t):         # vitk_cast_many reads 16-byte vectors: a parameter that is a view at an odd offset of a flat buffer is copied first
            t = t.detach().contiguous()
            return t.clone() if t.data_ptr() % 16 else t
        live = [(aligned(p), o) for p, o in zip(params, outs) if p.numel()]
        K.cast_many([a for a, _ in live], [b for _, b in live])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        outs = [None if g is None else torch.empty(g.shape, dtype=F32, device=g.device) for g in gs]
        live = [(g.contiguous() if g.data_ptr() % 16 == 0 else g.contiguous().clone(), o) for g, o in zip(gs, outs) if g is not None and g.numel()]
        if live:
            half = {g.dtype for g, _ in live}
            for hd in half:       # (one source dtype per call)
                K.cast_many([g for g, _ in live if g.dtype == hd], [o for g, o in live if g.dtype == hd])
        return (None, *outs)


def observed(m: nn.Module) -> bool:
    """Somebody registered a hook that expects to see this module run (forward, forward-pre, backward, backward-pre): the fused stages
    never call the module, so a model with such a hook inside a stage runs that stage op by op."""
    return bool(m._forward_hooks) or bool(m._forward_pre_hooks) or bool(m._backward_hooks) or bool(getattr(m, "_backward_pre_hooks", None))


def exactly(m, *classes) -> bool:
    """`m` is an instance of exactly one of these classes -- not of a subclass: a user's LoRALinear(nn.Linear) / QuantLinear / spectral-norm
    wrapper overrides forward, and a fused stage that read only `.weight` would silently drop what that forward adds."""
    return type(m) in classes


def plain_linear(m) -> bool:
    return exactly(m, Linear, nn.Linear) and not getattr(m, "parametrizations", None)


def plain_layernorm(m) -> bool:
    return exactly(m, LayerNorm, nn.LayerNorm) and m.elementwise_affine and m.weight is not None


def eager_modules(namespace) -> None:
    """Call at the end of a module file: every nn.Module defined there runs OUTSIDE torch.compile / TorchDynamo (`torch.compiler.disable`
    on its forward).  The fused stages are autograd Functions over ctypes calls into libvitk -- nothing Dynamo can trace (fake tensors
    have no device pointers; it raised InternalTorchDynamoError instead of breaking the graph) and nothing Inductor could improve: a
    user's `torch.compile(model)` (or a compiled model that contains one of these modules) runs these parts eagerly, the rest compiled."""
    disable = getattr(getattr(torch, "compiler", None), "disable", None)
    if disable is None:
        return
    for obj in list(namespace.values()):
        if isinstance(obj, type) and issubclass(obj, nn.Module) and obj.__module__ == namespace.get("__name__") and "forward" in obj.__dict__:
            obj.forward = disable(obj.__dict__["forward"])


def _autocast_dtype():
    """The 16-bit dtype of an active torch.autocast region on the GPU, else None."""
    try:
        if not torch.is_autocast_enabled("cuda"):
            return None
        d = torch.get_autocast_dtype("cuda")
    except TypeError:       # older torch: no device argument
        if not torch.is_autocast_enabled():
            return None
        d = torch.get_autocast_gpu_dtype()
    return d if d in (torch.bfloat16, torch.float16) else None


def _empty_batch_forward(call, module, v, args, kwargs):
    """A batch of ZERO images (the reference returns an empty (0, ...) result and zero gradients: vit.py:118-138 is shape-agnostic in the
    batch).  No kernel has anything to do; the result's trailing shape and dtype are read off a one-image stand-in run without autograd, and
    the empty result is tied to the input and the parameters so that backward() delivers (empty / zero) gradients like the reference's."""
    one = v.new_zeros((1,) + tuple(v.shape[1:]))
    was_training = module.training
    with torch.no_grad():
        if args:
            out = call(module, one, *args[1:], **kwargs)
        else:
            key = next(k for k in kwargs if kwargs[k] is v)
            out = call(module, **{**kwargs, key: one})
    assert module.training == was_training
    link = None
    if torch.is_grad_enabled():
        terms = [t for t in [v] + list(module.parameters()) if t.requires_grad]
        if terms:
            link = sum(t.float().sum() for t in terms) * 0.0

    def empty(o):
        if isinstance(o, torch.Tensor) and o.dim() >= 1 and o.shape[0] == 1:
            z = o.new_zeros((0,) + tuple(o.shape[1:]))
            return z if link is None or not z.is_floating_point() else z + link.to(z.dtype)
        if isinstance(o, tuple) and hasattr(o, "_fields"):
            return type(o)(*(empty(u) for u in o))
        if isinstance(o, (list, tuple)):
            return type(o)(empty(u) for u in o)
        return o

    return empty(out)


def autocast_aware(forward):
    """Decorator of a top-level model's forward.  The reference under `torch.autocast("cuda", dtype=torch.bfloat16)` -- or accelerate's
    mixed precision around train_vit_decorr.py:74 -- keeps float32 master parameters and runs its Linear layers on 16-bit copies.  The
    fused engine is not made of torch ops, so autocast cannot reach into it; instead, a float32 model called inside an autocast region
    runs the 16-bit engine on 16-bit copies of its parameters (CastParamsFn: gradients arrive in float32 on the master parameters, a
    GradScaler sees what it expects) and returns 16-bit logits like the reference's last Linear does.  Models that already are 16-bit,
    and calls outside autocast, go straight through."""
    import functools

    IMAGE_KEYS = ("img", "x", "images", "batched_images", "video")

    @functools.wraps(forward)
    def wrapped(self, *args, **kwargs):      # (every argument may come by keyword: model(img=t), a trainer's model(**batch))
        v = args[0] if args else next((kwargs[k] for k in IMAGE_KEYS if k in kwargs), None)
        if isinstance(v, torch.Tensor) and v.dim() >= 2 and v.shape[0] == 0:
            return _empty_batch_forward(wrapped, self, v, args, kwargs)
        with model_grad_scope():             # torch.no_grad() around the call is seen by every fused stage, not just the transformer
            return inner(self, *args, **kwargs)

    def inner(self, *args, **kwargs):
        dt = _autocast_dtype()
        if dt is None:
            return forward(self, *args, **kwargs)
        # remove_duplicate=False: every alias of a tied / shared parameter is swapped, not just the first name
        named = [(n, p) for n, p in self.named_parameters(remove_duplicate=False) if p.dtype == F32 and p.is_cuda]
        if not named:
            with torch.autocast("cuda", enabled=False):
                return forward(self, *args, **kwargs)
        uniq = {}
        for _, p in named:
            uniq.setdefault(id(p), p)
        cast_all = CastParamsFn.apply(dt, *uniq.values())
        for p_, c in zip(uniq.values(), cast_all):
            c._vitk_weight = True            # ops.is_weight: these get the K-blocked / transposed copies a Parameter gets
            c._vitk_master = (id(p_), weight_key(p_))      # fp8.Fp8State keys its e4m3 weight copies on the MASTER parameter's value
        by_id = dict(zip(uniq.keys(), cast_all))
        swap = {n: by_id[id(p)] for n, p in named}
        for n, b in self.named_buffers(remove_duplicate=False):
            if b.dtype == F32 and b.is_cuda and b.is_floating_point():
                swap[n] = _to(b, dt)

        # Only the IMAGE argument is cast (the first positional argument, or the keyword the reference's forwards call it by): under the
        # reference's torch.autocast every other tensor a wrapper passes along -- float32 targets, teacher logits, masks, temperatures --
        # keeps its dtype, and only autocast-eligible ops downcast.
        def conv(v):
            if isinstance(v, torch.Tensor):
                return cast(v, dt) if v.is_floating_point() else v
            if isinstance(v, tuple) and hasattr(v, "_fields"):       # namedtuple: positional constructor
                return type(v)(*(conv(u) for u in v))
            if isinstance(v, (list, tuple)):
                return type(v)(conv(u) for u in v)
            return v

        if args:
            args = (conv(args[0]),) + tuple(args[1:])
            kwargs = dict(kwargs)
        else:
            kwargs = {k: (conv(v) if k in IMAGE_KEYS else v) for k, v in kwargs.items()}
        # (the swap is in place for the duration of the call: concurrent forwards of ONE module from several threads would see each
        # other's 16-bit tensors -- like torch.func.functional_call, not thread-safe per module)
        # The reference under autocast keeps its residual stream in float32 (only Linear / matmul run in 16 bit): so does this route --
        # float32 forward and backward streams around the 16-bit GEMMs (ops.stream_policy) -- and lands at the reference-autocast's own
        # distance from float32 (tests/test_autocast_parity_gpu.py: 3.6e-3 on ViT-B/16 logits, where the pure-bf16 model is at 9.2e-3).
        # VITK_AUTOCAST_STREAM=16 keeps the parameter-dtype streams of a pure 16-bit model (faster by the two streams' bytes).
        pol = ops.stream_policy() if os.environ.get("VITK_AUTOCAST_STREAM", "f32") == "16" else ops.stream_policy(fwd16=False, grad16=False)
        with torch.autocast("cuda", enabled=False), pol:
            try:
                from torch.nn.utils.stateless import _reparametrize_module
            except ImportError:       # public route: re-enters __call__ (hooks on the top-level module then fire twice)
                return torch.func.functional_call(self, swap, args, kwargs, strict=False)
            with _reparametrize_module(self, swap, strict=False):
                return forward(self, *args, **kwargs)

    return wrapped


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        K.require_device(x, w)
        x = x.contiguous()
        rows, D = _rows(x)
        if w.dtype == F32 and x.dtype != F32:
            x = _to(x, F32)
        y = torch.empty(x.shape, dtype=w.dtype, device=x.device)
        mean, rstd = ops.ln_fwd(x, w, b, rows, D, y)
        ctx.save_for_backward(x, w, mean, rstd)
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, rstd = ctx.saved_tensors
        rows, D = _rows(x)
        T = w.dtype
        dy = _to(dy, T)
        dx32 = torch.empty(x.shape, dtype=F32, device=x.device)
        dw = torch.empty_like(w)
        db = torch.empty_like(w) if ctx.has_b else None
        dbuf = db if db is not None else torch.empty_like(w)
        ops.ln_bwd(dy, x, w, mean, rstd, rows, D, dx_f32=dx32, dw=dw, db=dbuf)
        return _to(dx32, x.dtype), dw, db


class LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        K.require_device(x, w)
        if x.dtype != w.dtype:
            raise VitkError(f"Linear: input dtype {x.dtype} != weight dtype {w.dtype}")
        x = x.contiguous()
        M, Kd = _rows(x)
        y = ops.linear_fwd(x, w, b, M)
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        M, Kd = _rows(x)
        dy = _to(dy, w.dtype)
        dw = torch.empty_like(w)
        db = torch.empty(w.shape[0], dtype=w.dtype, device=w.device) if ctx.has_b else None
        ops.linear_dw(dy, x, M, dw, db)
        dx = ops.linear_dx(dy, w, M).view(x.shape) if ctx.needs_input_grad[0] else None
        return dx, dw, db


class GELUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        K.require_device(x)
        x = x.contiguous()
        y = torch.empty_like(x)
        K.gelu_fwd(x, y)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        K.gelu_bwd(_to(dy, x.dtype), x, dx)
        return dx


class SoftmaxFn(torch.autograd.Function):
    """nn.Softmax(dim=-1) (vit.py:41,59)."""

    @staticmethod
    def forward(ctx, s):
        K.require_device(s)
        s = s.contiguous()
        rows, cols = _rows(s)
        p = torch.empty_like(s)
        K.softmax_fwd(s, p, rows, cols, 1.0)
        ctx.save_for_backward(p)
        return p

    @staticmethod
    def backward(ctx, dp):
        (p,) = ctx.saved_tensors
        rows, cols = _rows(p)
        ds = torch.empty_like(p)
        K.softmax_bwd(p, _to(dp, p.dtype), ds, rows, cols, 1.0)
        return ds


class AddFn(torch.autograd.Function):
    """a + b with both of one shape (the residual adds of vit.py:80-81)."""

    @staticmethod
    def forward(ctx, a, b):
        K.require_device(a, b)
        if a.shape != b.shape or a.dtype != b.dtype:
            raise VitkError("AddFn: operands must have one shape and dtype")
        a = a.contiguous(); b = b.contiguous()
        rows, D = _rows(a)
        out = torch.empty_like(a)
        K.add_rows(a, b, None, out, rows, D)
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g


class DropoutFn(torch.autograd.Function):
    """nn.Dropout in training mode (vit.py:22,24,42,48,109): counter-based mask, reproducible fwd<->bwd."""
    _offset = [0]

    @staticmethod
    def forward(ctx, x, p: float):
        K.require_device(x)
        x = x.contiguous()
        y = torch.empty_like(x)
        mask = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
        seed = (int(torch.initial_seed()) + 0x632BE5AB * dist_rank()) & 0x7fffffffffffffff
        off = DropoutFn._offset[0]
        DropoutFn._offset[0] += x.numel()
        K.dropout_fwd(x, y, mask, p, seed, off)
        ctx.save_for_backward(mask)
        ctx.p = p
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        K.dropout_bwd(dy, mask, dx, ctx.p)
        return dx, None


class PatchifyFn(torch.autograd.Function):
    """Rearrange 'b c (h p1) (w p2) -> b (h w) (p1 p2 c)' (vit.py:100)."""

    @staticmethod
    def forward(ctx, img, p1: int, p2: int):
        K.require_device(img)
        img = img.contiguous()
        B, C, H, W = img.shape
        if H % p1 or W % p2:
            raise VitkError("Image dimensions must be divisible by the patch size.")
        out = torch.empty((B, (H // p1) * (W // p2), p1 * p2 * C), dtype=img.dtype, device=img.device)
        K.patchify(img, out, B, C, H, W, p1, p2)
        ctx.geom = (B, C, H, W, p1, p2)
        return out

    @staticmethod
    def backward(ctx, g):
        if not ctx.needs_input_grad[0]:
            return None, None, None
        B, C, H, W, p1, p2 = ctx.geom              # the input itself requires a gradient (saliency, adversarial inputs): the inverse scatter
        g = g.contiguous()
        dimg = torch.empty((B, C, H, W), dtype=g.dtype, device=g.device)
        K.unpatchify(g, dimg, B, C, H, W, p1, p2)
        return dimg, None, None


class ScoresFn(torch.autograd.Function):
    """dots = (q k^T) * scale over the merged qkv tensor (vit.py:54-57): (B,N,3I) -> (B,H,N,N)."""

    @staticmethod
    def forward(ctx, qkv, heads: int, scale: float):
        K.require_device(qkv)
        qkv = qkv.contiguous()
        B, N, I3 = qkv.shape
        I = I3 // 3
        d = I // heads
        S = torch.empty((B, heads, N, N), dtype=qkv.dtype, device=qkv.device)
        K.gemm_generic(K.mat(qkv, I3, 1, N * I3, d), K.mat(qkv, 1, I3, N * I3, d, offset=I),
                       K.mat(S, N, 1, heads * N * N, N * N), N, N, d, nb1=B, nb2=heads, alpha=scale)
        ctx.save_for_backward(qkv)
        ctx.meta = (heads, scale)
        return S

    @staticmethod
    def backward(ctx, dS):
        (qkv,) = ctx.saved_tensors
        heads, scale = ctx.meta
        B, N, I3 = qkv.shape
        I = I3 // 3
        d = I // heads
        dS = _to(dS, qkv.dtype)
        dqkv = torch.zeros_like(qkv)  # the v third receives no gradient from here
        sm = K.mat(dS, N, 1, heads * N * N, N * N)
        smT = K.mat(dS, 1, N, heads * N * N, N * N)
        K.gemm_generic(sm, K.mat(qkv, I3, 1, N * I3, d, offset=I), K.mat(dqkv, I3, 1, N * I3, d), N, d, N,
                       nb1=B, nb2=heads, alpha=scale)
        K.gemm_generic(smT, K.mat(qkv, I3, 1, N * I3, d), K.mat(dqkv, I3, 1, N * I3, d, offset=I), N, d, N,
                       nb1=B, nb2=heads, alpha=scale)
        return dqkv, None, None


class AttnValuesFn(torch.autograd.Function):
    """out = attn @ v, merged back to 'b n (h d)' (vit.py:62-63): (B,H,N,N),(B,N,3I) -> (B,N,I)."""

    @staticmethod
    def forward(ctx, attn, qkv, heads: int):
        K.require_device(attn, qkv)
        attn = attn.contiguous(); qkv = qkv.contiguous()
        B, N, I3 = qkv.shape
        I = I3 // 3
        d = I // heads
        o = torch.empty((B, N, I), dtype=qkv.dtype, device=qkv.device)
        K.gemm_generic(K.mat(attn, N, 1, heads * N * N, N * N), K.mat(qkv, I3, 1, N * I3, d, offset=2 * I),
                       K.mat(o, I, 1, N * I, d), N, d, N, nb1=B, nb2=heads)
        ctx.save_for_backward(attn, qkv)
        ctx.heads = heads
        return o

    @staticmethod
    def backward(ctx, do):
        attn, qkv = ctx.saved_tensors
        heads = ctx.heads
        B, N, I3 = qkv.shape
        I = I3 // 3
        d = I // heads
        do = _to(do, qkv.dtype)
        dom = K.mat(do, I, 1, N * I, d)
        dattn = torch.empty_like(attn)
        K.gemm_generic(dom, K.mat(qkv, 1, I3, N * I3, d, offset=2 * I), K.mat(dattn, N, 1, heads * N * N, N * N),
                       N, N, d, nb1=B, nb2=heads)
        dqkv = torch.zeros_like(qkv)
        K.gemm_generic(K.mat(attn, 1, N, heads * N * N, N * N), dom, K.mat(dqkv, I3, 1, N * I3, d, offset=2 * I),
                       N, d, N, nb1=B, nb2=heads)
        return dattn, dqkv, None


class FusedAttnFn(torch.autograd.Function):
    """The fused attention core on the merged qkv tensor: (B,N,3I) -> (B,N,I)."""

    @staticmethod
    def forward(ctx, qkv, heads: int, scale: float):
        K.require_device(qkv)
        qkv = qkv.contiguous()
        B, N, I3 = qkv.shape
        d = I3 // 3 // heads
        o, saved = ops.attn_fwd(qkv.view(B * N, I3), B, N, heads, d, scale)
        ctx.save_for_backward(qkv, o, saved)
        ctx.meta = (heads, scale)
        return o.view(B, N, I3 // 3)

    @staticmethod
    def backward(ctx, do):
        qkv, o, saved = ctx.saved_tensors
        heads, scale = ctx.meta
        B, N, I3 = qkv.shape
        d = I3 // 3 // heads
        dqkv = ops.attn_bwd(qkv.view(B * N, I3), o, _to(do, qkv.dtype).view(B * N, I3 // 3), saved, B, N, heads, d, scale)
        return dqkv.view(B, N, I3), None, None


class MeanTokensFn(torch.autograd.Function):
    """x.mean(dim=1) (vit.py:135, simple_vit.py:117)."""

    @staticmethod
    def forward(ctx, x):
        K.require_device(x)
        x = x.contiguous()
        B, N, D = x.shape
        out = torch.empty((B, D), dtype=x.dtype, device=x.device)
        K.mean_pool_fwd(x, out, B, N, D)
        ctx.shape = (B, N, D)
        return out

    @staticmethod
    def backward(ctx, g):
        B, N, D = ctx.shape
        g = g.contiguous()
        dx = torch.empty((B, N, D), dtype=g.dtype, device=g.device)
        K.mean_pool_bwd(g, dx, B, N, D)
        return dx


# ---- nn.Module shells (same classes / state_dict keys as the reference's children) -------------
class ConcatTokensFn(torch.autograd.Function):
    """torch.cat((front tokens broadcast over the batch, x), dim=1) + pos[:N]: the cls token + positional table of vit.py:122-127,
    the register tokens of simple_vit_with_register_tokens.py:113-115 (front / pos may be None).  One launch for all images."""

    @staticmethod
    def forward(ctx, x, front, pos, behind):
        """behind: the extra tokens FOLLOW x -- pack([x, r]) of simple_vit_with_register_tokens.py:113-115 -- instead of leading it."""
        K.require_device(x)
        x = x.contiguous()
        B, Np, D = x.shape
        F_ = 0 if front is None else front.shape[0]
        N = Np + F_
        fr = None if front is None or F_ == 0 else _to(front.reshape(F_, D), x.dtype)
        ps = None if pos is None else _to(pos[:N].contiguous(), x.dtype)
        out = torch.empty((B, N, D), dtype=x.dtype, device=x.device)
        K.concat_tokens(x, fr, ps, out, B, Np, (-F_ if behind else F_) if fr is not None else 0, D)
        ctx.meta = (B, Np, F_, D, None if front is None else (front.shape, front.dtype), None if pos is None else (pos.shape, pos.dtype))
        ctx.behind = bool(behind)
        return out

    @staticmethod
    def backward(ctx, g):
        B, Np, F_, D, front_meta, pos_meta = ctx.meta
        N = Np + F_
        g = g.contiguous()
        dx = torch.empty((B, Np, D), dtype=g.dtype, device=g.device)
        x0 = 0 if ctx.behind else F_                                                            # first row of x inside every image
        K.copy_cols(g.view(B, N * D)[:, x0 * D:], N * D, dx, Np * D, B, Np * D, Np * D)
        dfront = dpos = None
        if front_meta is not None or pos_meta is not None:
            gsum = torch.empty((N, D), dtype=g.dtype, device=g.device)
            ops.colsum(g, B, N * D, gsum)                                                     # sum over the batch
            if front_meta is not None:
                gfr = gsum[Np:] if ctx.behind else gsum[:F_]
                dfront = _to(gfr.contiguous(), front_meta[1]).reshape(front_meta[0]) if F_ else torch.zeros(front_meta[0], dtype=front_meta[1], device=g.device)
            if pos_meta is not None:
                dpos = torch.zeros(pos_meta[0], dtype=pos_meta[1], device=g.device)
                K.cast(gsum, dpos[:N])
        return dx, dfront, dpos, None


class TokenSliceFn(torch.autograd.Function):
    """x[:, start:stop] as a contiguous tensor (unpack of the register tokens: simple_vit_with_register_tokens.py:119)."""

    @staticmethod
    def forward(ctx, x, start: int, stop=None):
        x = x.contiguous()
        B, N, D = x.shape
        stop = N if stop is None else stop
        n = stop - start
        out = torch.empty((B, n, D), dtype=x.dtype, device=x.device)
        K.copy_cols(x.view(B, N * D)[:, start * D:], N * D, out, n * D, B, n * D, n * D)
        ctx.meta = (B, N, D, start, n)
        return out

    @staticmethod
    def backward(ctx, g):
        B, N, D, start, n = ctx.meta
        g = g.contiguous()
        dx = torch.zeros((B, N, D), dtype=g.dtype, device=g.device)
        K.copy_cols(g, n * D, dx.view(B, N * D)[:, start * D:], N * D, B, n * D, n * D)
        return dx, None, None


class GatherTokensFn(torch.autograd.Function):
    """x[batch_indices, keep] (vit_with_patch_dropout.py:32): keep is int32 (B, Kp) with distinct entries per image."""

    @staticmethod
    def forward(ctx, x, keep):
        K.require_device(x)
        x = x.contiguous()
        B, Np, D = x.shape
        keep = keep.to(torch.int32).contiguous()
        Kp = keep.shape[1]
        out = torch.empty((B, Kp, D), dtype=x.dtype, device=x.device)
        K.gather_tokens(x, keep, out, B, Np, Kp, D)
        ctx.keep = keep
        ctx.meta = (B, Np, Kp, D)
        return out

    @staticmethod
    def backward(ctx, g):
        B, Np, Kp, D = ctx.meta
        g = g.contiguous()
        dx = torch.zeros((B, Np, D), dtype=g.dtype, device=g.device)
        K.gather_tokens(g, ctx.keep, dx, B, Np, Kp, D, scatter=True)
        return dx, None


class LayerNorm(nn.LayerNorm):
    def forward(self, x):
        return LayerNormFn.apply(x, self.weight, self.bias)


class Linear(nn.Linear):
    def forward(self, x):
        return LinearFn.apply(x, self.weight, self.bias)


class GELU(nn.GELU):
    def forward(self, x):
        return GELUFn.apply(x)


class Softmax(nn.Softmax):
    def forward(self, x):
        assert self.dim in (-1, x.dim() - 1)
        return SoftmaxFn.apply(x)


class Dropout(nn.Dropout):
    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        return DropoutFn.apply(x, float(self.p))


class Patchify(nn.Module):
    """Stands where the reference has einops' Rearrange (to_patch_embedding[0]); no parameters."""

    def __init__(self, p1: int, p2: int):
        super().__init__()
        self.p1, self.p2 = p1, p2

    def forward(self, img):
        return PatchifyFn.apply(img, self.p1, self.p2)

    def extra_repr(self):
        return f"'b c (h p1) (w p2) -> b (h w) (p1 p2 c)', p1={self.p1}, p2={self.p2}"


eager_modules(globals())
