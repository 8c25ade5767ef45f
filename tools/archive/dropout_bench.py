"""ViT-B/16 forward+backward with active dropout (vit.py:22,24,42,48,109 at p = 0.1) vs p = 0.  python tools/dropout_bench.py"""
import time
import torch
from vit_pytorch_amd import ViT

dev = "cuda"
for p in (0.0, 0.1):
    torch.manual_seed(0)
    m = ViT(image_size=224, patch_size=16, num_classes=1000, dim=768, depth=12, heads=12, mlp_dim=3072, dropout=p, emb_dropout=p).to(dev, dtype=torch.bfloat16).train()
    x = torch.randn(256, 3, 224, 224, device=dev).to(torch.bfloat16)
    y = torch.randint(0, 1000, (256,), device=dev)
    def step():
        m.zero_grad(set_to_none=True)
        torch.nn.functional.cross_entropy(m(x).float(), y).backward()
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"dropout {p}: {dt * 1e3:.1f} ms/step, {256 / dt:.0f} img/s, peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    del m
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
