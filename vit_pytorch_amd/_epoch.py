"""Generation counter of the parameter VALUES, for caches derived from weights (transposed copies for the dX GEMMs, e4m3
copies for the fp8 forward).  torch bumps `tensor._version` on in-place torch ops, but this package's fused optimizer
(`optim.Adam.step` -> `vitk_adam_step`) writes parameters through raw pointers, which torch cannot see -- it bumps this
counter instead, and every cache keys on (data_ptr, _version, weights_epoch).

torch's own FUSED optimizers are in the same position: `torch.optim.AdamW(..., fused=True)` (`torch._fused_adamw_`) updates the
parameters WITHOUT touching their version counters ([measured] torch 2.10: `_version` 4 -> 4 across a step, foreach / single-tensor
flavours 2 per step) -- found by tests/test_train_loop_gpu.py: the K-blocked weight copies kept serving the initial weights and the
model silently stopped learning through the fused GEMMs.  So a global optimizer-step post hook (every torch.optim.Optimizer subclass,
whatever its implementation) bumps the epoch too."""
_EPOCH = [0]


def weights_epoch() -> int:
    return _EPOCH[0]


def bump_weights_epoch() -> None:
    _EPOCH[0] += 1


_NOCACHE = [0]


def weight_key(w):
    import os
    if os.environ.get("VITK_WEIGHT_CACHE", "1") == "0":       # never equal to a stored key: every use re-derives the copies
        _NOCACHE[0] += 1
        return (w.data_ptr(), w._version, _EPOCH[0], tuple(w.shape), w.dtype, _NOCACHE[0])
    return (w.data_ptr(), w._version, _EPOCH[0], tuple(w.shape), w.dtype)


def _install_optimizer_hook() -> None:
    try:
        from torch.optim.optimizer import register_optimizer_step_post_hook
    except ImportError:       # a torch without global optimizer hooks: version counters and invalidate_weight_caches() remain
        return
    register_optimizer_step_post_hook(lambda optimizer, args, kwargs: bump_weights_epoch())


_install_optimizer_hook()


def invalidate_weight_caches() -> None:
    """Call after writing parameter VALUES behind torch's back -- `p.data.mul_()`, `p.data.copy_()`, `dist.broadcast(p.data)`, a raw
    pointer write: `tensor.data` views do not bump `p._version`, so the caches keyed on it (K-blocked / transposed / split / e4m3
    weight copies) would keep serving the old values.  `load_state_dict`, every torch.optim optimizer (a global step hook bumps the
    epoch: the fused flavours do not touch version counters) and every in-place op ON THE PARAMETER ITSELF are seen without it.  `VITK_WEIGHT_CACHE=0` disables the caches altogether (every call re-derives its copies: debugging)."""
    bump_weights_epoch()


# ---- grad mode of the CALLER of a fused autograd.Function -------------------------------------------------------------
# Inside autograd.Function.forward torch.is_grad_enabled() is always False and ctx.needs_input_grad mirrors requires_grad even under
# torch.no_grad(), so the module-level caller records the mode here (thread-local) right before .apply().
import threading as _threading

_TLS = _threading.local()


def note_grad_mode(enabled: bool) -> None:
    _TLS.grad = bool(enabled)


def caller_grad_mode() -> bool:
    """The one-shot note of the module that is about to call a fused Function, AND the mode the enclosing model-level forward was
    entered in (model_grad_scope): the stages in front of / behind the transformer (patch embedding, head) run before the note is set /
    after it is reset, and under torch.no_grad() they must not build and cache the backward's transposed weight packs."""
    return getattr(_TLS, "grad", True) and getattr(_TLS, "model", True)


class model_grad_scope:
    """with model_grad_scope(): ... around a model-level forward (functional.autocast_aware opens it): records torch.is_grad_enabled()
    for every fused stage the forward runs; nests (the previous value comes back)."""

    def __enter__(self):
        import torch
        self.prev = getattr(_TLS, "model", True)
        _TLS.model = self.prev and torch.is_grad_enabled()
        return self

    def __exit__(self, *exc):
        _TLS.model = self.prev
        return False


def reset_grad_mode() -> None:
    """Called by the fused Functions at the end of their forward: the note is ONE-SHOT.  A later .apply() that nobody announced (a direct
    caller, a future module) then sees the default -- grad enabled: activations are kept, backward works -- instead of whatever an
    earlier no_grad forward left behind (which would make backward crash on an empty ctx.saved)."""
    _TLS.grad = True
