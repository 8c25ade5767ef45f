"""Per-kernel TF/s of the fp8 GEMM flavours next to their bf16 counterparts at BASELINE config 5's extents (ViT-H/14 @ 336,
M = batch x 577 token rows; default batch 64) -- python tools/fp8_kernels_bench.py [batch]

    NT forward  e4m3 x e4m3     K = 32 form / K = 128 form / bf16 (persistent kernel)
    NT dX       e5m2 x e4m3     K = 32 form / K = 128 form / bf16
    TN dW       e5m2^T x e4m3   K = 32 form / K = 128 form / bf16 (split-M kernel)
    the one-pass delayed quantiser (GB/s)

One line per measurement, JSON at the end (tools: not part of the product path)."""
import json
import sys

import torch

sys.path.insert(0, "tools")
from kbench import timeit  # noqa: E402
from vit_pytorch_amd import _lib as L, kernels as K  # noqa: E402

dev, BF = "cuda", torch.bfloat16
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
M, D, I, F = batch * 577, 1280, 1280, 5120
res = {}


def q(x, fmt):
    fmax = 448.0 if fmt == K.FMT_E4M3 else 57344.0
    sc = torch.empty(2, device=dev)
    a = x.float().abs().max()
    sc[0] = fmax / a; sc[1] = a / fmax
    x8 = torch.empty(x.shape, dtype=torch.uint8, device=dev)
    K.quantize_fp8_delayed(x, x8, sc, None, fmt)
    return x8, sc


def report(name, ms, flops):
    tf = flops / ms / 1e9
    res[name] = {"ms": round(ms, 4), "tflops": round(tf, 1)}
    print(f"{name:44s} {ms:8.3f} ms  {tf:8.1f} TF/s", flush=True)


for name, (n, k, epi) in {"qkv": (3 * I, D, L.EPI_NONE), "ff1+gelu": (F, D, L.EPI_BIAS_GELU), "ff2+resid": (D, F, L.EPI_RESID)}.items():
    A = torch.randn(M, k, device=dev).to(BF); W = (torch.randn(n, k, device=dev) * k ** -0.5).to(BF)
    bias = torch.randn(n, device=dev).to(BF)
    A8, sa = q(A, K.FMT_E4M3); W8, sw = q(W, K.FMT_E4M3)
    if epi == L.EPI_RESID:
        C = torch.zeros(M, n, device=dev); kw = dict(bias=bias, resid=C)
    elif epi == L.EPI_BIAS_GELU:
        C = torch.empty(M, n, dtype=BF, device=dev); kw = dict(bias=bias, aux=torch.empty(M, n, dtype=BF, device=dev))
    else:
        C = torch.empty(M, n, dtype=BF, device=dev); kw = {}
    fl = 2.0 * M * n * k
    report(f"NT fwd {name} bf16", timeit(lambda: K.gemm_nt_bf16(A, k, W, k, C, n, M, n, k, epi, **kw), iters=5, warm=2), fl)
    for k128 in (False, True):
        report(f"NT fwd {name} e4m3 K={'128' if k128 else '32'}",
               timeit(lambda: K.gemm_nt_fp8_v2(A8, k, W8, k, C, n, M, n, k, epi, a_kind=K.A_E4M3, alpha_a=sa[1:], alpha_w=sw[1:], k128=k128, **kw), iters=5, warm=2), fl)
    del A, W, A8, W8, C, kw

for name, (nw, kd, gelu) in {"dff2 (GELU')": (D, F, True), "dff1": (F, D, False), "dqkv": (3 * I, D, False)}.items():
    dY = (torch.randn(M, nw, device=dev) * 1e-3).to(BF); Wt = (torch.randn(kd, nw, device=dev) * nw ** -0.5).to(BF)
    dY8, sa = q(dY, K.FMT_E5M2); W8, sw = q(Wt, K.FMT_E4M3)
    C = torch.empty(M, kd, dtype=BF, device=dev)
    pre = torch.randn(M, kd, device=dev).to(BF) if gelu else None
    epi = L.EPI_GELU_BWD if gelu else L.EPI_NONE
    fl = 2.0 * M * nw * kd
    report(f"NT dX {name} bf16", timeit(lambda: K.gemm_nt_bf16(dY, nw, Wt, nw, C, kd, M, kd, nw, epi, aux=pre), iters=5, warm=2), fl)
    for k128 in (False, True):
        report(f"NT dX {name} e5m2 K={'128' if k128 else '32'}",
               timeit(lambda: K.gemm_nt_fp8_v2(dY8, nw, W8, nw, C, kd, M, kd, nw, epi, a_kind=K.A_E5M2, aux=pre, alpha_a=sa[1:], alpha_w=sw[1:], k128=k128), iters=5, warm=2), fl)
    del dY, Wt, dY8, W8, C, pre

for name, (n, kd) in {"dW2": (D, F), "dW1": (F, D), "dWqkv": (3 * I, D), "dWout": (D, I)}.items():
    dY = (torch.randn(M, n, device=dev) * 1e-3).to(BF); X = torch.randn(M, kd, device=dev).to(BF)
    dY8, sy = q(dY, K.FMT_E5M2); X8, sx = q(X, K.FMT_E4M3)
    dW = torch.empty(n, kd, dtype=BF, device=dev)
    fl = 2.0 * M * n * kd
    s16 = K.gemm_tn_splits(M, n, kd)
    ws = torch.empty(max(s16, 64) * n * kd, device=dev)
    report(f"TN {name} bf16 ({s16} splits)", timeit(lambda: K.gemm_tn_bf16(dY, n, X, kd, dW, kd, M, n, kd, ws, s16), iters=5, warm=2), fl)
    for k128 in (False, True):
        s8 = K.gemm_tn_fp8_splits(M, n, kd, k128)
        report(f"TN {name} e5m2 x e4m3 K={'128' if k128 else '32'} ({s8} splits)",
               timeit(lambda: K.gemm_tn_fp8(dY8, n, X8, kd, dW, kd, M, n, kd, ws, s8, alpha_y=sy[1:], alpha_x=sx[1:], k128=k128), iters=5, warm=2), fl)
    del dY, X, dY8, X8, dW, ws

x = torch.randn(M, F, device=dev).to(BF)
x8 = torch.empty(M, F, dtype=torch.uint8, device=dev)
sc = torch.tensor([100.0, 0.01], device=dev); am = torch.zeros(64, dtype=torch.int32, device=dev)
ms = timeit(lambda: K.quantize_fp8_delayed(x, x8, sc, am, K.FMT_E5M2), iters=5, warm=2)
res["quantize_fp8_delayed"] = {"ms": round(ms, 4), "gbps": round(3.0 * M * F / ms / 1e6, 1)}
print(f"quantize_fp8_delayed (M x {F}: 3 B / element)      {ms:8.3f} ms  {3.0 * M * F / ms / 1e6:8.1f} GB/s")
print(json.dumps({"batch": batch, "M": M, "results": res}))
