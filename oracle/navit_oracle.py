"""CPU oracle for the NaViT path (BASELINE config 4).

TEST INFRASTRUCTURE ONLY (see oracle/vit_oracle.py header for the rules).

Restates /root/reference/vit_pytorch/na_vit.py in the reference's OWN padded / dense-mask formulation
(so that it is an independent check of the product's packed, mask-free formulation):
LayerNorm without bias :82-89, RMSNorm :93-101, Attention with q/k norm and boolean attn_mask,
scale = 1 :115-169, Transformer :171-193, NaViT.forward :255-402 (patch order 'c p1 p2' :300,
factorised positional embedding :354-359, attention pooling :371-387, real-image selection :393-396).
Token dropout (:306-314) draws from torch's RNG and is excluded from parity (token_dropout_prob=None).

Pinned to outputs of the reference itself: tests/golden/navit_two_packs.npz written by
oracle/make_golden.py; checked in tests/test_oracle.py.
"""
from __future__ import annotations

from typing import Dict, List

import torch

from . import vit_oracle as O

Tensor = torch.Tensor


def rms_norm_heads(x: Tensor, gamma: Tensor) -> Tensor:
    """na_vit.py:93-101: F.normalize(x, dim=-1) * sqrt(d) * gamma, x (b,h,n,d), gamma (h,1,d)."""
    n = x.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    return x / n * (x.shape[-1] ** 0.5) * gamma


def masked_attention(q: Tensor, k: Tensor, v: Tensor, mask: Tensor) -> Tensor:
    """F.scaled_dot_product_attention(q,k,v, attn_mask=bool mask, scale=1.) (na_vit.py:161-166);
    fully masked rows give 0 (torch >= 2.5 CPU behaviour; such rows never reach an output)."""
    s = q @ k.transpose(-1, -2)
    s = s.masked_fill(~mask, float("-inf"))
    p = torch.softmax(s, dim=-1)
    p = torch.nan_to_num(p, nan=0.0)
    return p @ v


def attention(x: Tensor, p: Dict[str, Tensor], prefix: str, heads: int, mask: Tensor, context: Tensor = None) -> Tensor:
    xn, _, _ = O.layer_norm_fwd(x, p[prefix + "norm.gamma"], None)
    kv_in = context if context is not None else xn
    q = O.linear_fwd(xn, p[prefix + "to_q.weight"], None)
    kv = O.linear_fwd(kv_in, p[prefix + "to_kv.weight"], None)
    I = q.shape[-1]
    k, v = kv[..., :I], kv[..., I:]
    q, k, v = (O.split_heads(t, heads) for t in (q, k, v))
    q = rms_norm_heads(q, p[prefix + "q_norm.gamma"])
    k = rms_norm_heads(k, p[prefix + "k_norm.gamma"])
    out = O.merge_heads(masked_attention(q, k, v, mask))
    return O.linear_fwd(out, p[prefix + "to_out.0.weight"], None)


def feed_forward(x: Tensor, p: Dict[str, Tensor], prefix: str) -> Tensor:
    xn, _, _ = O.layer_norm_fwd(x, p[prefix + "0.gamma"], None)
    h = O.gelu_fwd(O.linear_fwd(xn, p[prefix + "1.weight"], p[prefix + "1.bias"]))
    return O.linear_fwd(h, p[prefix + "4.weight"], p[prefix + "4.bias"])


def navit_fwd(batched_images: List[List[Tensor]], p: Dict[str, Tensor], *, patch_size: int, depth: int, heads: int) -> Tensor:
    ps = patch_size
    dtype = p["pos_embed_height"].dtype
    seqs, poss, ids, nimgs = [], [], [], []
    for images in batched_images:
        nimgs.append(len(images))
        s_list, p_list, i_list = [], [], []
        for i, img in enumerate(images):
            c, H, W = img.shape
            ph, pw = H // ps, W // ps
            x = img.to(dtype).reshape(c, ph, ps, pw, ps).permute(1, 3, 0, 2, 4).reshape(ph * pw, c * ps * ps)  # (h w) (c p1 p2)
            s_list.append(x)
            hh = torch.arange(ph).repeat_interleave(pw); ww = torch.arange(pw).repeat(ph)
            p_list.append(torch.stack([hh, ww], -1))
            i_list.append(torch.full((ph * pw,), i, dtype=torch.long))
        seqs.append(torch.cat(s_list)); poss.append(torch.cat(p_list)); ids.append(torch.cat(i_list))
    b = len(seqs)
    lengths = torch.tensor([s.shape[0] for s in seqs])
    n = int(lengths.max())
    pad = lambda ts, fill=0: torch.stack([torch.cat([t, t.new_full((n - t.shape[0],) + t.shape[1:], fill)]) for t in ts])
    patches, positions, image_ids = pad(seqs), pad(poss), pad(ids)
    key_pad = torch.arange(n)[None, :] < lengths[:, None]
    attn_mask = (image_ids[:, None, :, None] == image_ids[:, None, None, :]) & key_pad[:, None, None, :]

    x, _, _ = O.layer_norm_fwd(patches, p["to_patch_embedding.0.gamma"], None)
    x = O.linear_fwd(x, p["to_patch_embedding.1.weight"], p["to_patch_embedding.1.bias"])
    x, _, _ = O.layer_norm_fwd(x, p["to_patch_embedding.2.gamma"], None)
    x = x + p["pos_embed_height"][positions[..., 0]] + p["pos_embed_width"][positions[..., 1]]

    for i in range(depth):
        x = attention(x, p, f"transformer.layers.{i}.0.", heads, attn_mask) + x
        x = feed_forward(x, p, f"transformer.layers.{i}.1.") + x
    x, _, _ = O.layer_norm_fwd(x, p["transformer.norm.gamma"], None)

    maxq = max(nimgs)
    queries = p["attn_pool_queries"][None, None, :].expand(b, maxq, -1)
    pool_mask = (torch.arange(maxq)[None, :, None] == image_ids[:, None, :]) & key_pad[:, None, :]
    x = attention(queries, p, "attn_pool.", heads, pool_mask[:, None], context=x) + queries
    x = x.reshape(b * maxq, -1)
    is_img = (torch.arange(maxq)[None, :] < torch.tensor(nimgs)[:, None]).reshape(-1)
    x = x[is_img]
    x, _, _ = O.layer_norm_fwd(x, p["mlp_head.0.gamma"], None)
    return O.linear_fwd(x, p["mlp_head.1.weight"], None)


def run_fwd_bwd(cfg: dict, params: Dict[str, Tensor], batched_images, dtype=torch.float32):
    p = {k: v.detach().to(dtype).clone().requires_grad_(v.is_floating_point() and not k.endswith(".beta")) for k, v in params.items()}
    imgs = [[im.to(dtype) for im in g] for g in batched_images]
    out = navit_fwd(imgs, p, patch_size=cfg["patch_size"], depth=cfg["depth"], heads=cfg["heads"])
    O.loss_fn(out).backward()
    grads = {k: (v.grad.detach() if v.grad is not None else torch.zeros_like(v)) for k, v in p.items() if not k.endswith(".beta")}
    return out.detach(), grads
