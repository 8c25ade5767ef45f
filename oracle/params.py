"""Deterministic, torch-RNG-independent weights and inputs for parity tests.

TEST INFRASTRUCTURE ONLY (see oracle/vit_oracle.py header).

The reference constructors draw weights from torch's global RNG, which cannot be
replayed on the GPU box (the reference is absent there).  Parity fixtures
therefore use weights generated HERE from a numpy ``default_rng(seed)`` stream
and loaded into the reference with ``load_state_dict`` by ``make_golden.py``.
The key names and shapes are the ``state_dict`` contract of the reference
(vit.py:99-116, simple_vit.py:90-108); distributions
mimic the reference's defaults (Linear: U(-1/sqrt(fan_in), 1/sqrt(fan_in));
cls/pos: N(0,1)) except that LayerNorm affine parameters are perturbed away
from (1, 0) so that their gradients and the affine path are exercised.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np
import torch


def _pair(t):
    return t if isinstance(t, tuple) else (t, t)


def param_shapes(kind: str, cfg: dict) -> "OrderedDict[str, Tuple[int, ...]]":
    """state_dict key -> shape, in the reference's registration order."""
    ih, iw = _pair(cfg["image_size"])
    ph, pw = _pair(cfg["patch_size"])
    ch = cfg.get("channels", 3)
    D, depth, heads = cfg["dim"], cfg["depth"], cfg["heads"]
    dh = cfg.get("dim_head", 64)
    F = cfg["mlp_dim"]
    C = cfg["num_classes"]
    I = heads * dh
    P = ch * ph * pw
    Np = (ih // ph) * (iw // pw)
    simple = kind == "simple_vit"
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    if not simple:
        ncls = 1 if cfg.get("pool", "cls") == "cls" else 0
        s["cls_token"] = (ncls, D)
        s["pos_embedding"] = (Np + ncls, D)
    s["to_patch_embedding.1.weight"] = (P,)
    s["to_patch_embedding.1.bias"] = (P,)
    s["to_patch_embedding.2.weight"] = (D, P)
    s["to_patch_embedding.2.bias"] = (D,)
    s["to_patch_embedding.3.weight"] = (D,)
    s["to_patch_embedding.3.bias"] = (D,)
    s["transformer.norm.weight"] = (D,)
    s["transformer.norm.bias"] = (D,)
    for i in range(depth):
        a = f"transformer.layers.{i}.0."
        f = f"transformer.layers.{i}.1."
        s[a + "norm.weight"] = (D,)
        s[a + "norm.bias"] = (D,)
        s[a + "to_qkv.weight"] = (3 * I, D)
        if simple:
            s[a + "to_out.weight"] = (D, I)
        elif not (heads == 1 and dh == D):
            s[a + "to_out.0.weight"] = (D, I)
            s[a + "to_out.0.bias"] = (D,)
        s[f + "net.0.weight"] = (D,)
        s[f + "net.0.bias"] = (D,)
        s[f + "net.1.weight"] = (F, D)
        s[f + "net.1.bias"] = (F,)
        second = "net.3." if simple else "net.4."
        s[f + second + "weight"] = (D, F)
        s[f + second + "bias"] = (D,)
    if simple:
        s["linear_head.weight"] = (C, D)
        s["linear_head.bias"] = (C,)
    elif C > 0:
        s["mlp_head.weight"] = (C, D)
        s["mlp_head.bias"] = (C,)
    return s


def make_params(kind: str, cfg: dict, seed: int) -> Dict[str, torch.Tensor]:
    rng = np.random.default_rng(seed)
    out: Dict[str, torch.Tensor] = OrderedDict()
    for name, shape in param_shapes(kind, cfg).items():
        n = int(np.prod(shape)) if len(shape) else 1
        if name in ("cls_token", "pos_embedding"):
            a = rng.standard_normal(n)
        elif len(shape) == 2:  # Linear weight (out, in)
            bound = 1.0 / np.sqrt(shape[1])
            a = rng.uniform(-bound, bound, n)
        elif name.endswith("norm.weight") or name.endswith("net.0.weight") or \
                name in ("to_patch_embedding.1.weight", "to_patch_embedding.3.weight"):
            a = 1.0 + 0.1 * rng.standard_normal(n)  # LayerNorm gamma
        elif name.endswith("norm.bias") or name.endswith("net.0.bias") or \
                name in ("to_patch_embedding.1.bias", "to_patch_embedding.3.bias"):
            a = 0.1 * rng.standard_normal(n)  # LayerNorm beta
        else:  # Linear bias
            a = rng.uniform(-0.05, 0.05, n)
        out[name] = torch.from_numpy(np.asarray(a, dtype=np.float32).reshape(shape)).clone()
    return out


def make_images(cfg: dict, batch: int, seed: int, image_size=None) -> torch.Tensor:
    ih, iw = _pair(image_size if image_size is not None else cfg["image_size"])
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((batch, cfg.get("channels", 3), ih, iw)).astype(np.float32)
    return torch.from_numpy(a)


# The parity cases.  cfg1 is BASELINE.json configs[0].
CASES = {
    "cfg1_simple_vit_tiny": dict(
        kind="simple_vit", batch=8, seed=0,
        cfg=dict(image_size=32, patch_size=4, num_classes=10, dim=64, depth=2, heads=4, mlp_dim=128)),
    "vit_cls_tiny": dict(
        kind="vit", batch=3, seed=1,
        cfg=dict(image_size=32, patch_size=8, num_classes=7, dim=32, depth=2, heads=2, dim_head=16,
                 mlp_dim=64, pool="cls")),
    "vit_mean_rect": dict(  # non-square image/patch, mean pool, dim_head != dim/heads
        kind="vit", batch=2, seed=2,
        cfg=dict(image_size=(24, 32), patch_size=(4, 8), num_classes=5, dim=48, depth=1, heads=3,
                 dim_head=8, mlp_dim=40, pool="mean")),
    "vit_identity_out": dict(  # heads == 1 and dim_head == dim: to_out is nn.Identity, no out-projection parameters (vit.py:34,49)
        kind="vit", batch=3, seed=4,
        cfg=dict(image_size=32, patch_size=8, num_classes=5, dim=64, depth=2, heads=1, dim_head=64, mlp_dim=96, pool="cls")),
    "vit_dim30": dict(  # a width that is not a multiple of 4 (missing 5): the any-width kernels, op by op
        kind="vit", batch=3, seed=6,
        cfg=dict(image_size=32, patch_size=8, num_classes=5, dim=30, depth=2, heads=2, dim_head=16, mlp_dim=50, pool="cls")),
    "vit_tokens_small_input": dict(  # num_classes=0 -> tokens out; input smaller than image_size (vit.py:125-127)
        kind="vit", batch=2, seed=3, image=(16, 24),
        cfg=dict(image_size=32, patch_size=8, num_classes=0, dim=32, depth=1, heads=2, dim_head=16,
                 mlp_dim=48, pool="cls")),
}


# Cases at the LAYER SHAPES of BASELINE configs 2, 3 and 5 (depth cut to 2 / 1 so the reference finishes on the CPU in seconds,
# batches sized so that M = batch * N >= 1024: the production NT GEMM (persistent kernel) and the whole-head / chunked attention
# kernels are what the GPU runs).  Their goldens are COMPACT (oracle/make_golden.py::main_wide): full logits, and per parameter
# gradient its L2 norm plus the elements at `sample_index(numel)`.
WIDE_CASES = {
    "vit_b16_width": dict(kind="vit", batch=6, seed=11,
                          cfg=dict(image_size=224, patch_size=16, num_classes=1000, dim=768, depth=2, heads=12, mlp_dim=3072)),
    "simple_vit_b16_width": dict(kind="simple_vit", batch=6, seed=12,
                                 cfg=dict(image_size=224, patch_size=16, num_classes=1000, dim=768, depth=2, heads=12, mlp_dim=3072)),
    "vit_l16_width": dict(kind="vit", batch=6, seed=13,
                          cfg=dict(image_size=224, patch_size=16, num_classes=1000, dim=1024, depth=2, heads=16, mlp_dim=4096)),
    "vit_h14_width": dict(kind="vit", batch=2, seed=14,
                          cfg=dict(image_size=336, patch_size=14, num_classes=1000, dim=1280, depth=1, heads=16, dim_head=80, mlp_dim=5120)),
    # FULL DEPTH (round 3): BASELINE config 2 (ViT-B/16, 12 layers) and config 3 (ViT-L/16, 24 layers) exactly as benchmarked, at a
    # batch the reference finishes on the CPU (M = 6 * 197 = 1,182 >= 1,024: production kernels).  1,024 samples per gradient.
    "vit_b16_full": dict(kind="vit", batch=6, seed=21, sample=1024,
                         cfg=dict(image_size=224, patch_size=16, num_classes=1000, dim=768, depth=12, heads=12, mlp_dim=3072)),
    "vit_l16_full": dict(kind="vit", batch=6, seed=22, sample=1024,
                         cfg=dict(image_size=224, patch_size=16, num_classes=1000, dim=1024, depth=24, heads=16, mlp_dim=4096)),
    # THE HEADLINE RUN ITSELF (round 4): BASELINE config 2 at its own batch, 256 -- M = 50,432 token rows, i.e. the tile plans, split
    # counts and grid shapes bench.py times, end to end against the reference (f32 and its own bf16 run).  1,024 samples per gradient.
    "vit_b16_full_b256": dict(kind="vit", batch=256, seed=23, sample=1024,
                              cfg=dict(image_size=224, patch_size=16, num_classes=1000, dim=768, depth=12, heads=12, mlp_dim=3072)),
    # BASELINE config 5's architecture at depth 4 / batch 8 (M = 4,616): held against the reference with the activation-recompute policy
    # FORCED ON (what config 5 takes at batch 256), so that path answers to the reference and not to the repo's own full-save run.
    "vit_h14_d4_b8": dict(kind="vit", batch=8, seed=24, sample=1024,
                          cfg=dict(image_size=336, patch_size=14, num_classes=1000, dim=1280, depth=4, heads=16, dim_head=80, mlp_dim=5120)),
    # BASELINE config 5's architecture at its FULL depth of 32 (round 6), at a batch the reference finishes on the CPU (M = 8 * 577 = 4,616):
    # the fp8 and bf16 engines are held to the reference through all 32 layers, not 4.  512 samples per gradient.
    "vit_h14_full_b8": dict(kind="vit", batch=8, seed=25, sample=512,
                            cfg=dict(image_size=336, patch_size=14, num_classes=1000, dim=1280, depth=32, heads=16, dim_head=80, mlp_dim=5120)),
}
GOLD_SAMPLE = 4096


def sample_index(numel: int, sample: int = GOLD_SAMPLE) -> np.ndarray:
    """The (at most `sample`) flat positions of a gradient tensor that a compact golden stores: a fixed stride walk."""
    if numel <= sample:
        return np.arange(numel, dtype=np.int64)
    return (np.arange(sample, dtype=np.int64) * 7919) % numel


# ---- sibling variants: simple_vit_with_qk_norm.py, simple_vit_with_register_tokens.py, vit_with_patch_dropout.py
VARIANT_CASES = {
    "qk_norm_tiny": dict(module="simple_vit_with_qk_norm", cls="SimpleViT", batch=3, seed=31,
                         cfg=dict(image_size=32, patch_size=8, num_classes=7, dim=64, depth=2, heads=2, dim_head=64, mlp_dim=96)),
    "register_tokens_tiny": dict(module="simple_vit_with_register_tokens", cls="SimpleViT", batch=3, seed=32,
                                 cfg=dict(image_size=32, patch_size=8, num_classes=7, dim=64, depth=2, heads=2, dim_head=64, mlp_dim=96,
                                          num_register_tokens=4)),
    "patch_dropout_eval_tiny": dict(module="vit_with_patch_dropout", cls="ViT", batch=3, seed=33,
                                    cfg=dict(image_size=32, patch_size=8, num_classes=7, dim=64, depth=2, heads=2, dim_head=64, mlp_dim=96,
                                             pool="cls", patch_dropout=0.25)),
}


def make_params_for(shapes: "OrderedDict[str, Tuple[int, ...]]", seed: int) -> Dict[str, torch.Tensor]:
    """Deterministic values for any state_dict given its (key, shape) list -- same value rules as make_params: 2-D tensors are
    Linear weights U(-1/sqrt(fan_in), ..), LayerNorm weights 1 + 0.1 N(0,1) and biases 0.1 N(0,1), Linear biases U(-0.05, 0.05),
    token / positional tables N(0,1), q/k norm gammas (heads, 1, d) around their 1/sqrt(d) initial value."""
    rng = np.random.default_rng(seed)
    out: Dict[str, torch.Tensor] = OrderedDict()
    for name, shape in shapes.items():
        n = int(np.prod(shape)) if len(shape) else 1
        last = name.split(".")[-1]
        if last == "gamma" and len(shape) == 3:
            a = (1.0 + 0.1 * rng.standard_normal(n)) / np.sqrt(shape[-1])
        elif "token" in name or "pos_embedding" in name:
            a = rng.standard_normal(n)
        elif len(shape) == 2:
            bound = 1.0 / np.sqrt(shape[1])
            a = rng.uniform(-bound, bound, n)
        elif last == "weight":                       # 1-D weight: a LayerNorm gamma
            a = 1.0 + 0.1 * rng.standard_normal(n)
        elif last == "bias" and ("norm" in name or name.endswith("net.0.bias")):
            a = 0.1 * rng.standard_normal(n)         # LayerNorm beta
        else:
            a = rng.uniform(-0.05, 0.05, n)
        out[name] = torch.from_numpy(np.asarray(a, dtype=np.float32).reshape(shape)).clone()
    return out


# ---- NaViT (BASELINE config 4) --------------------------------------------------------------------------------
def navit_param_shapes(cfg: dict) -> "OrderedDict[str, Tuple[int, ...]]":
    """state_dict key -> shape of na_vit.NaViT in registration order (buffers `beta` included)."""
    ih, iw = _pair(cfg["image_size"])
    ps = cfg["patch_size"]
    ch = cfg.get("channels", 3)
    D, depth, heads, F, C = cfg["dim"], cfg["depth"], cfg["heads"], cfg["mlp_dim"], cfg["num_classes"]
    dh = cfg.get("dim_head", 64)
    I = heads * dh
    P = ch * ps * ps
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    s["pos_embed_height"] = (ih // ps, D)
    s["pos_embed_width"] = (iw // ps, D)
    s["attn_pool_queries"] = (D,)
    s["to_patch_embedding.0.gamma"] = (P,); s["to_patch_embedding.0.beta"] = (P,)
    s["to_patch_embedding.1.weight"] = (D, P); s["to_patch_embedding.1.bias"] = (D,)
    s["to_patch_embedding.2.gamma"] = (D,); s["to_patch_embedding.2.beta"] = (D,)

    def attn(prefix):
        s[prefix + "norm.gamma"] = (D,); s[prefix + "norm.beta"] = (D,)
        s[prefix + "q_norm.gamma"] = (heads, 1, dh)
        s[prefix + "k_norm.gamma"] = (heads, 1, dh)
        s[prefix + "to_q.weight"] = (I, D)
        s[prefix + "to_kv.weight"] = (2 * I, D)
        s[prefix + "to_out.0.weight"] = (D, I)

    for i in range(depth):
        attn(f"transformer.layers.{i}.0.")
        f = f"transformer.layers.{i}.1."
        s[f + "0.gamma"] = (D,); s[f + "0.beta"] = (D,)
        s[f + "1.weight"] = (F, D); s[f + "1.bias"] = (F,)
        s[f + "4.weight"] = (D, F); s[f + "4.bias"] = (D,)
    s["transformer.norm.gamma"] = (D,); s["transformer.norm.beta"] = (D,)
    attn("attn_pool.")
    s["mlp_head.0.gamma"] = (D,); s["mlp_head.0.beta"] = (D,)
    s["mlp_head.1.weight"] = (C, D)
    return s


def make_navit_params(cfg: dict, seed: int) -> Dict[str, torch.Tensor]:
    rng = np.random.default_rng(seed)
    out: Dict[str, torch.Tensor] = OrderedDict()
    for name, shape in navit_param_shapes(cfg).items():
        n = int(np.prod(shape))
        if name.endswith(".beta"):
            a = np.zeros(n)                                   # registered buffer, always zero (na_vit.py:86)
        elif name.startswith("pos_embed") or name == "attn_pool_queries":
            a = rng.standard_normal(n)
        elif name.endswith("gamma"):
            a = 1.0 + 0.1 * rng.standard_normal(n)
        elif len(shape) == 2:
            bound = 1.0 / np.sqrt(shape[1])
            a = rng.uniform(-bound, bound, n)
        else:
            a = rng.uniform(-0.05, 0.05, n)
        out[name] = torch.from_numpy(np.asarray(a, dtype=np.float32).reshape(shape)).clone()
    return out


def make_navit_images(cfg: dict, sizes, seed: int):
    """sizes: list (packs) of lists of (H, W).  Returns nested lists of (C, H, W) float32 images."""
    rng = np.random.default_rng(seed)
    ch = cfg.get("channels", 3)
    return [[torch.from_numpy(rng.standard_normal((ch, h, w)).astype(np.float32)) for (h, w) in pack] for pack in sizes]


NAVIT_CASES = {
    "navit_two_packs": dict(
        seed=5, sizes=[[(32, 48), (16, 16), (64, 24)], [(40, 40), (8, 56)]],
        cfg=dict(image_size=64, patch_size=8, num_classes=7, dim=64, depth=2, heads=2, mlp_dim=96)),
}

# BASELINE config 4 at its real width (dim 1024, 16 heads, mlp 4096, patch 16), depth 2: ONE pack of exactly 4,096 tokens from 32
# images of mixed resolutions (256, 128, 64, 16 and 32 patches).  Compact golden (logits + norms and samples of every gradient),
# reference in f32 and the reference's own bf16 run (make_golden.main_navit_wide).
def navit_bench_sizes():
    """The image sizes of bench.py --config navit (BASELINE config 4's bench workload): a*16 x b*16 px with a, b ~ U{4..40} from
    numpy default_rng(0) until >= 32,768 tokens: 65 images, grouped into 9 packs of <= 4,096 tokens by group_images (na_vit.py:38-77)."""
    import numpy as np
    rng = np.random.default_rng(0)
    sizes, tok = [], 0
    while tok < 32768:
        a, b = rng.integers(4, 41, 2)
        sizes.append((int(a) * 16, int(b) * 16)); tok += int(a) * int(b)
    return sizes


# The BENCH workload of config 4 (the 65-image / 9-pack draw of bench.py --config navit, ~33 k tokens) through the reference at config 4's
# width -- at depth 12, half of the depth the bench times: the reference's masked attention keeps (packs, heads, 4096, 4096) score tensors for
# the backward, 44 GiB peak at depth 12 in f32 on this container's 62 GiB (depth 16 did not fit).  It pins what is specific to the bench
# workload -- the greedy grouping of the flat list, nine ragged packs at once, per-image attention pooling, factorised positions at sizes up to
# 640 px -- at a depth where 16-bit error has accumulated over twelve layers; the remaining depth is pinned by the 24-layer ViT-L/16 golden on
# the same kernels.
NAVIT_BENCH_CASES = {
    "navit_bench_draw_d12": dict(
        seed=43, sample=1024, flat=True, group_max_seq_len=4096,
        cfg=dict(image_size=1024, patch_size=16, num_classes=1000, dim=1024, depth=12, heads=16, mlp_dim=4096)),
}

NAVIT_WIDE_CASES = {
    "navit_cfg4_width": dict(
        seed=41, sample=1024,
        sizes=[[(256, 256), (128, 128), (256, 128), (64, 64), (256, 256), (128, 64), (256, 256), (64, 64), (256, 256), (128, 128),
                (256, 256), (64, 64), (256, 128), (256, 256), (128, 64), (64, 64), (256, 256), (128, 128), (256, 256), (64, 64),
                (256, 128), (256, 256), (128, 64), (64, 64), (256, 256), (128, 128), (256, 256), (64, 64), (256, 128), (256, 256),
                (128, 64), (64, 64)]],
        cfg=dict(image_size=256, patch_size=16, num_classes=1000, dim=1024, depth=2, heads=16, mlp_dim=4096)),
    # BASELINE config 4's architecture at its FULL depth of 24 (round 6): one pack of 2,048 tokens (16 images of four resolutions) -- what the
    # reference's dense masked attention lets the host hold at 24 layers (the bench draw's 9 packs of 4,096 tokens stop at depth 12).
    "navit_cfg4_full_depth": dict(
        seed=42, sample=512, tokens=2048,
        sizes=[[(256, 256), (128, 128), (256, 128), (64, 64), (256, 256), (128, 64), (256, 256), (64, 64), (256, 256), (128, 128),
                (256, 256), (64, 64), (256, 128), (256, 256), (128, 64), (64, 64)]],
        cfg=dict(image_size=256, patch_size=16, num_classes=1000, dim=1024, depth=24, heads=16, mlp_dim=4096)),
}

