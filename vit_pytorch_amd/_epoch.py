"""Generation counter of the parameter VALUES, for caches derived from weights (transposed copies for the dX GEMMs, e4m3
copies for the fp8 forward).  torch bumps `tensor._version` on in-place torch ops, but this package's fused optimizer
(`optim.Adam.step` -> `vitk_adam_step`) writes parameters through raw pointers, which torch cannot see -- it bumps this
counter instead, and every cache keys on (data_ptr, _version, weights_epoch)."""
_EPOCH = [0]


def weights_epoch() -> int:
    return _EPOCH[0]


def bump_weights_epoch() -> None:
    _EPOCH[0] += 1


def weight_key(w):
    return (w.data_ptr(), w._version, _EPOCH[0], tuple(w.shape), w.dtype)
