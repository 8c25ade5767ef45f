"""Serving-size latency of the ViT-B/16 forward: eager launch path vs one HIP-graph replay.  python tools/serve_bench.py"""
import time
import torch
from vit_pytorch_amd import ViT
from vit_pytorch_amd.graphs import GraphedForward

dev = "cuda"
torch.manual_seed(0)
m = ViT(image_size=224, patch_size=16, num_classes=1000, dim=768, depth=12, heads=12, mlp_dim=3072).to(dev, dtype=torch.bfloat16).eval()


def timed(fn, n=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for B in (1, 4, 16, 64):
    x = torch.randn(B, 3, 224, 224, device=dev).to(torch.bfloat16)
    with torch.no_grad():
        te = timed(lambda: m(x))
    g = GraphedForward(m, x)
    tg = timed(lambda: g(x))
    print(f"ViT-B/16 forward, batch {B:3d}: eager {te * 1e3:7.3f} ms ({B / te:8.0f} img/s)   graph {tg * 1e3:7.3f} ms ({B / tg:8.0f} img/s)   x{te / tg:.2f}")
