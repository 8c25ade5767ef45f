"""CPU oracle for the ViT / SimpleViT encoder forward + backward hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``vit_pytorch_amd/`` may import this
file; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` do, and there only as the checker.

What this is
------------
A functional restatement, in elementary tensor arithmetic on the CPU, of the
algorithm in the reference files

    /root/reference/vit_pytorch/vit.py          (FeedForward :15-28, Attention :30-64,
                                                 Transformer :66-83, ViT :85-138)
    /root/reference/vit_pytorch/simple_vit.py   (posemb_sincos_2d :12-21, FeedForward :25-35,
                                                 Attention :37-62, Transformer :64-78,
                                                 SimpleViT :80-120)

The reference itself is pure Python over ``torch.nn`` / ``einops``; the
arithmetic lives in PyTorch ATen (third-party, ``torch>=2.4`` per the
reference's pyproject.toml:35; this image has torch 2.10.0).  So every function
below restates the *published definition* of the ATen op the reference calls
(LayerNorm: biased variance, eps inside the sqrt; GELU: exact erf form; softmax
over the last axis; Linear: x @ W^T + b) with nothing but + - * / sqrt exp erf
matmul and reshapes, and cites the reference call site it stands for.

Two halves:

* ``*_fwd`` functions: the forward algorithm.  Differentiable by autograd, so
  ``loss.backward()`` on them gives oracle gradients.
* ``*_bwd`` functions: the explicit backward formulas that the HIP kernels
  implement (LayerNorm bwd, GELU bwd, softmax/SDPA bwd, Linear dX/dW/db).  They
  are checked against autograd of the forward in ``tests/test_oracle.py``; they
  exist so a kernel can be checked stage-by-stage, not only end-to-end.

Pinning
-------
The reference's own tests hold NO numerical vectors for this path
(tests/test_vit.py:20 asserts a shape only), so "golden vectors from the
reference's tests" do not exist.  The oracle is pinned instead against outputs of
the reference itself executed in the build container:
``oracle/make_golden.py`` imports ``/root/reference/vit_pytorch/{vit,simple_vit}.py``
by file path, runs forward+backward on deterministic weights/inputs
(``oracle/params.py``) and stores logits + every parameter gradient under
``tests/golden/*.npz``; ``tests/test_oracle.py`` requires this restatement to
reproduce them (fp32, rel-L2 <= 2e-6; fp64 oracle as arbiter).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch

Tensor = torch.Tensor
LN_EPS = 1e-5  # nn.LayerNorm default, used at vit.py:19,39,69,101,103


# --------------------------------------------------------------------------- #
# elementary ops (forward)
# --------------------------------------------------------------------------- #
def layer_norm_fwd(x: Tensor, w: Tensor, b: Optional[Tensor], eps: float = LN_EPS):
    """nn.LayerNorm over the last axis (vit.py:19,39,69,101,103).

    Biased variance, eps added inside the square root, affine.  Returns
    (y, mean, rstd) -- mean/rstd are what the HIP forward saves for backward.
    """
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(var + eps)
    y = (x - mean) * rstd * w
    if b is not None:
        y = y + b
    return y, mean.squeeze(-1), rstd.squeeze(-1)


def gelu_fwd(x: Tensor) -> Tensor:
    """nn.GELU() default = exact erf form (vit.py:21, simple_vit.py:31)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def linear_fwd(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    """nn.Linear: y = x @ W^T + b, W stored (out, in) (vit.py:20,23,44,47,102,116)."""
    y = x @ w.transpose(-1, -2)
    if b is not None:
        y = y + b
    return y


def softmax_fwd(s: Tensor) -> Tensor:
    """nn.Softmax(dim=-1) (vit.py:41,59)."""
    m = s.max(dim=-1, keepdim=True).values
    e = torch.exp(s - m)
    return e / e.sum(dim=-1, keepdim=True)


def patchify(img: Tensor, p1: int, p2: int) -> Tensor:
    """Rearrange 'b c (h p1) (w p2) -> b (h w) (p1 p2 c)' (vit.py:100, simple_vit.py:91).

    The patch vector is channel-fastest: element ((i*p2)+j)*c + ch is pixel
    (row i, col j) of the patch, channel ch.
    """
    b, c, H, W = img.shape
    h, w = H // p1, W // p2
    x = img.reshape(b, c, h, p1, w, p2)
    x = x.permute(0, 2, 4, 3, 5, 1)  # b h w p1 p2 c
    return x.reshape(b, h * w, p1 * p2 * c)


def posemb_sincos_2d(h: int, w: int, dim: int, temperature: float = 10000.0) -> Tensor:
    """simple_vit.py:12-21.  [sin(x w), cos(x w), sin(y w), cos(y w)], x = column index."""
    assert dim % 4 == 0, "feature dimension must be multiple of 4 for sincos emb"
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    omega = torch.arange(dim // 4) / (dim // 4 - 1)
    omega = 1.0 / (temperature ** omega)
    y = ys.flatten()[:, None] * omega[None, :]
    x = xs.flatten()[:, None] * omega[None, :]
    return torch.cat((x.sin(), x.cos(), y.sin(), y.cos()), dim=1).to(torch.float32)


def split_heads(t: Tensor, heads: int) -> Tensor:
    """rearrange 'b n (h d) -> b h n d' (vit.py:55): h is the OUTER factor."""
    b, n, hd = t.shape
    return t.reshape(b, n, heads, hd // heads).permute(0, 2, 1, 3)


def merge_heads(t: Tensor) -> Tensor:
    """rearrange 'b h n d -> b n (h d)' (vit.py:63)."""
    b, h, n, d = t.shape
    return t.permute(0, 2, 1, 3).reshape(b, n, h * d)


def attention_core_fwd(q: Tensor, k: Tensor, v: Tensor, scale: float):
    """vit.py:57-62: dots = (q k^T) * scale (scale AFTER the matmul); softmax; attn v."""
    dots = (q @ k.transpose(-1, -2)) * scale
    attn = softmax_fwd(dots)
    return attn @ v, attn


def attention_fwd(x: Tensor, p: Dict[str, Tensor], prefix: str, heads: int, dim_head: int,
                  simple: bool) -> Tensor:
    """Attention.forward (vit.py:51-64 / simple_vit.py:50-62)."""
    xn, _, _ = layer_norm_fwd(x, p[prefix + "norm.weight"], p[prefix + "norm.bias"])
    qkv = linear_fwd(xn, p[prefix + "to_qkv.weight"], None)  # bias=False, vit.py:44
    inner = heads * dim_head
    q, k, v = qkv[..., :inner], qkv[..., inner:2 * inner], qkv[..., 2 * inner:]  # chunk(3), q,k,v order
    q, k, v = (split_heads(t, heads) for t in (q, k, v))
    out, _ = attention_core_fwd(q, k, v, dim_head ** -0.5)
    out = merge_heads(out)
    if simple:
        return linear_fwd(out, p[prefix + "to_out.weight"], None)  # simple_vit.py:48 bias-free
    if (prefix + "to_out.0.weight") in p:
        return linear_fwd(out, p[prefix + "to_out.0.weight"], p[prefix + "to_out.0.bias"])
    return out  # nn.Identity when heads == 1 and dim_head == dim (vit.py:34,49)


def feed_forward_fwd(x: Tensor, p: Dict[str, Tensor], prefix: str, simple: bool) -> Tensor:
    """FeedForward.forward (vit.py:18-28 / simple_vit.py:28-35). Dropout p=0 -> identity."""
    second = "net.3." if simple else "net.4."
    xn, _, _ = layer_norm_fwd(x, p[prefix + "net.0.weight"], p[prefix + "net.0.bias"])
    h = gelu_fwd(linear_fwd(xn, p[prefix + "net.1.weight"], p[prefix + "net.1.bias"]))
    return linear_fwd(h, p[prefix + second + "weight"], p[prefix + second + "bias"])


def transformer_fwd(x: Tensor, p: Dict[str, Tensor], depth: int, heads: int, dim_head: int,
                    simple: bool, prefix: str = "transformer.") -> Tensor:
    """Transformer.forward (vit.py:78-83): x = attn(x) + x; x = ff(x) + x; final LayerNorm."""
    for i in range(depth):
        x = attention_fwd(x, p, f"{prefix}layers.{i}.0.", heads, dim_head, simple) + x
        x = feed_forward_fwd(x, p, f"{prefix}layers.{i}.1.", simple) + x
    y, _, _ = layer_norm_fwd(x, p[prefix + "norm.weight"], p[prefix + "norm.bias"])
    return y


def patch_embed_fwd(img: Tensor, p: Dict[str, Tensor], patch: Tuple[int, int]) -> Tensor:
    """to_patch_embedding (vit.py:99-104): patchify -> LN(patch_dim) -> Linear -> LN(dim)."""
    x = patchify(img, *patch)
    x, _, _ = layer_norm_fwd(x, p["to_patch_embedding.1.weight"], p["to_patch_embedding.1.bias"])
    x = linear_fwd(x, p["to_patch_embedding.2.weight"], p["to_patch_embedding.2.bias"])
    x, _, _ = layer_norm_fwd(x, p["to_patch_embedding.3.weight"], p["to_patch_embedding.3.bias"])
    return x


def vit_fwd(img: Tensor, p: Dict[str, Tensor], *, patch_size, depth: int, heads: int,
            dim_head: int = 64, pool: str = "cls", num_classes: int = 1000) -> Tensor:
    """ViT.forward (vit.py:118-138) with dropout = emb_dropout = 0."""
    patch = patch_size if isinstance(patch_size, tuple) else (patch_size, patch_size)
    x = patch_embed_fwd(img, p, patch)
    b = x.shape[0]
    cls = p["cls_token"]  # (1, D) for pool='cls', (0, D) for pool='mean' (vit.py:97,106)
    x = torch.cat((cls.unsqueeze(0).expand(b, -1, -1), x), dim=1)  # vit.py:122-123
    seq = x.shape[1]
    x = x + p["pos_embedding"][:seq]  # vit.py:125-127
    x = transformer_fwd(x, p, depth, heads, dim_head, simple=False)
    if num_classes == 0:
        return x  # vit.py:132-133
    x = x.mean(dim=1) if pool == "mean" else x[:, 0]  # vit.py:135
    return linear_fwd(x, p["mlp_head.weight"], p["mlp_head.bias"])  # vit.py:137-138


def simple_vit_fwd(img: Tensor, p: Dict[str, Tensor], *, patch_size, depth: int, heads: int,
                   dim_head: int = 64) -> Tensor:
    """SimpleViT.forward (simple_vit.py:110-120)."""
    patch = patch_size if isinstance(patch_size, tuple) else (patch_size, patch_size)
    x = patch_embed_fwd(img, p, patch)
    H, W = img.shape[-2:]
    pe = posemb_sincos_2d(H // patch[0], W // patch[1], x.shape[-1]).to(x.dtype)
    x = x + pe  # simple_vit.py:114 (in-place add of the sincos table cast to x.dtype)
    x = transformer_fwd(x, p, depth, heads, dim_head, simple=True)
    x = x.mean(dim=1)  # simple_vit.py:117
    return linear_fwd(x, p["linear_head.weight"], p["linear_head.bias"])


# --------------------------------------------------------------------------- #
# explicit backward formulas (what the HIP backward kernels implement)
# --------------------------------------------------------------------------- #
def layer_norm_bwd(dy: Tensor, x: Tensor, w: Tensor, mean: Tensor, rstd: Tensor):
    """Backward of layer_norm_fwd.  Rows = all leading axes flattened.

    xhat = (x-mean)*rstd ; g = dy*w
    dx   = rstd * (g - mean_j(g) - xhat * mean_j(g*xhat))
    dw   = sum_rows dy*xhat ; db = sum_rows dy
    """
    D = x.shape[-1]
    xhat = (x - mean.unsqueeze(-1)) * rstd.unsqueeze(-1)
    g = dy * w
    c1 = g.mean(dim=-1, keepdim=True)
    c2 = (g * xhat).mean(dim=-1, keepdim=True)
    dx = rstd.unsqueeze(-1) * (g - c1 - xhat * c2)
    dw = (dy * xhat).reshape(-1, D).sum(0)
    db = dy.reshape(-1, D).sum(0)
    return dx, dw, db


def gelu_bwd(dy: Tensor, x: Tensor) -> Tensor:
    """d/dx [0.5 x (1+erf(x/sqrt2))] = Phi(x) + x * phi(x)."""
    cdf = 0.5 * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))
    pdf = torch.exp(-0.5 * x * x) * (1.0 / math.sqrt(2.0 * math.pi))
    return dy * (cdf + x * pdf)


def linear_bwd(dy: Tensor, x: Tensor, w: Tensor, has_bias: bool):
    """dX = dY W ; dW = dY^T X (reduced over all rows) ; db = colsum(dY)."""
    N, K = w.shape
    dx = dy @ w
    dw = dy.reshape(-1, N).transpose(0, 1) @ x.reshape(-1, K)
    db = dy.reshape(-1, N).sum(0) if has_bias else None
    return dx, dw, db


def attention_core_bwd(do: Tensor, q: Tensor, k: Tensor, v: Tensor, scale: float):
    """Backward of attention_core_fwd in the flash form the HIP kernels use.

    P = softmax(scale * q k^T) recomputed from lse = logsumexp(scale * q k^T)
    delta_i = sum_d dO[i,d] * O[i,d]
    dV = P^T dO ; dP = dO V^T ; dS = P * (dP - delta) ; dQ = scale * dS K ; dK = scale * dS^T Q
    """
    s = (q @ k.transpose(-1, -2)) * scale
    lse = torch.logsumexp(s, dim=-1, keepdim=True)
    p = torch.exp(s - lse)
    o = p @ v
    delta = (do * o).sum(dim=-1, keepdim=True)
    dv = p.transpose(-1, -2) @ do
    dp = do @ v.transpose(-1, -2)
    ds = p * (dp - delta)
    dq = (ds @ k) * scale
    dk = (ds.transpose(-1, -2) @ q) * scale
    return dq, dk, dv


# --------------------------------------------------------------------------- #
# convenience: run fwd + bwd and collect grads
# --------------------------------------------------------------------------- #
def loss_fn(logits: Tensor) -> Tensor:
    """The scalar the parity tests differentiate: mean of squares of the output (fp32+)."""
    return logits.double().square().mean() if logits.dtype == torch.float64 else logits.float().square().mean()


def run_fwd_bwd(kind: str, cfg: dict, params: Dict[str, Tensor], img: Tensor, dtype=torch.float32):
    """Returns (logits, {name: grad}) from the oracle at the given dtype on the CPU."""
    p = {k: v.detach().to(dtype).clone().requires_grad_(v.numel() > 0) for k, v in params.items()}
    img = img.detach().to(dtype)
    if kind == "vit":
        out = vit_fwd(img, p, patch_size=cfg["patch_size"], depth=cfg["depth"], heads=cfg["heads"],
                      dim_head=cfg.get("dim_head", 64), pool=cfg.get("pool", "cls"),
                      num_classes=cfg["num_classes"])
    elif kind == "simple_vit":
        out = simple_vit_fwd(img, p, patch_size=cfg["patch_size"], depth=cfg["depth"],
                             heads=cfg["heads"], dim_head=cfg.get("dim_head", 64))
    else:
        raise ValueError(kind)
    loss_fn(out).backward()
    grads = {k: (v.grad.detach() if v.grad is not None else torch.zeros_like(v)) for k, v in p.items()}
    return out.detach(), grads
