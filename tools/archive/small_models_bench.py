"""Forward+backward throughput of the smaller standard ViTs (same kernels, narrower GEMMs).  python tools/small_models_bench.py"""
import time
import torch
from vit_pytorch_amd import ViT

dev = "cuda"
CFGS = {"ViT-Ti/16": dict(dim=192, depth=12, heads=3, mlp_dim=768), "ViT-S/16": dict(dim=384, depth=12, heads=6, mlp_dim=1536),
        "ViT-B/16": dict(dim=768, depth=12, heads=12, mlp_dim=3072), "ViT-B/32": dict(dim=768, depth=12, heads=12, mlp_dim=3072, patch_size=32)}
for name, c in CFGS.items():
    c = dict(image_size=224, patch_size=16, num_classes=1000, **c) if "patch_size" not in c else dict(image_size=224, num_classes=1000, **c)
    torch.manual_seed(0)
    m = ViT(**c).to(dev, dtype=torch.bfloat16)
    B = 256
    x = torch.randn(B, 3, 224, 224, device=dev).to(torch.bfloat16); y = torch.randint(0, 1000, (B,), device=dev)
    def step():
        m.zero_grad(set_to_none=True)
        torch.nn.functional.cross_entropy(m(x).float(), y).backward()
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    N = (224 // c["patch_size"]) ** 2 + 1; D = c["dim"]; F = c["mlp_dim"]; I = c["heads"] * 64
    fwd = 2 * (N - 1) * 3 * c["patch_size"] ** 2 * D + c["depth"] * (2 * N * D * 3 * I + 4 * c["heads"] * N * N * 64 + 2 * N * I * D + 4 * N * D * F) + 2 * D * 1000
    print(f"{name:9s} batch {B}: {dt * 1e3:6.1f} ms/step {B / dt:8.0f} img/s  {3 * fwd * B / dt / 1e12:6.1f} TFLOP/s ({3 * fwd * B / dt / 2.5166e15 * 100:4.1f} % of peak)")
    del m
