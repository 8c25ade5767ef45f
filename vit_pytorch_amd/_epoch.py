"""Generation counter of the parameter VALUES, for caches derived from weights (transposed copies for the dX GEMMs, e4m3
copies for the fp8 forward).  torch bumps `tensor._version` on in-place torch ops, but this package's fused optimizer
(`optim.Adam.step` -> `vitk_adam_step`) writes parameters through raw pointers, which torch cannot see -- it bumps this
counter instead, and every cache keys on (data_ptr, _version, weights_epoch)."""
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


def invalidate_weight_caches() -> None:
    """Call after writing parameter VALUES behind torch's back -- `p.data.mul_()`, `p.data.copy_()`, `dist.broadcast(p.data)`, a raw
    pointer write: `tensor.data` views do not bump `p._version`, so the caches keyed on it (K-blocked / transposed / split / e4m3
    weight copies) would keep serving the old values.  `load_state_dict`, optimizers and every in-place op ON THE PARAMETER ITSELF
    are seen without it.  `VITK_WEIGHT_CACHE=0` disables the caches altogether (every call re-derives its copies: debugging)."""
    bump_weights_epoch()


# ---- grad mode of the CALLER of a fused autograd.Function -------------------------------------------------------------
# Inside autograd.Function.forward torch.is_grad_enabled() is always False and ctx.needs_input_grad mirrors requires_grad even under
# torch.no_grad(), so the module-level caller records the mode here (thread-local) right before .apply().
import threading as _threading

_TLS = _threading.local()


def note_grad_mode(enabled: bool) -> None:
    _TLS.grad = bool(enabled)


def caller_grad_mode() -> bool:
    return getattr(_TLS, "grad", True)


def reset_grad_mode() -> None:
    """Called by the fused Functions at the end of their forward: the note is ONE-SHOT.  A later .apply() that nobody announced (a direct
    caller, a future module) then sees the default -- grad enabled: activations are kept, backward works -- instead of whatever an
    earlier no_grad forward left behind (which would make backward crash on an empty ctx.saved)."""
    _TLS.grad = True
