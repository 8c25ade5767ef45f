"""vit_pytorch_amd: MI355X-native drop-in for vit_pytorch.ViT / vit_pytorch.SimpleViT.

Mirrors the export surface of the reference package (vit_pytorch/__init__.py:1-5 exports
ViT and SimpleViT; MAE and Dino are out of scope).  Sub-modules: `na_vit` (NaViT on packed tokens),
`parallel` (flat-buffer data parallelism over RCCL), `optim` (fused Adam / AdamW), `graphs` (HIP-graph capture).
"""
__version__ = "0.1.0"


def __getattr__(name):
    # lazy: importing the package must not require torch.cuda or the built library
    if name in ("ViT", "Transformer", "Attention", "FeedForward"):
        from . import vit as _v
        return getattr(_v, name)
    if name == "SimpleViT":
        from .simple_vit import SimpleViT
        return SimpleViT
    if name == "invalidate_weight_caches":   # after raw `.data` writes to parameters (see _epoch.py)
        from ._epoch import invalidate_weight_caches
        return invalidate_weight_caches
    if name == "NaViT":   # reference: `from vit_pytorch.na_vit import NaViT` (README.md:154)
        from .na_vit import NaViT
        return NaViT
    raise AttributeError(name)
