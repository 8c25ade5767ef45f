"""Forward+backward step with 16-bit vs e4m3 forward GEMM operands (fp8.enable_fp8_forward).  python tools/fp8_bench.py"""
import time
import torch
from vit_pytorch_amd import ViT
from vit_pytorch_amd.fp8 import enable_fp8_forward

dev = "cuda"
CFGS = {"ViT-B/16 b256": (dict(image_size=224, patch_size=16, num_classes=1000, dim=768, depth=12, heads=12, mlp_dim=3072), 256),
        "ViT-H/14@336 b64": (dict(image_size=336, patch_size=14, num_classes=1000, dim=1280, depth=32, heads=16, dim_head=80, mlp_dim=5120), 64)}
for name, (c, B) in CFGS.items():
    res = {}
    for mode in ("bf16", "fp8-forward"):
        torch.manual_seed(0)
        m = ViT(**c).to(dev, dtype=torch.bfloat16)
        if mode != "bf16":
            enable_fp8_forward(m)
        x = torch.randn(B, 3, c["image_size"], c["image_size"], device=dev).to(torch.bfloat16); y = torch.randint(0, 1000, (B,), device=dev)
        def step():
            m.zero_grad(set_to_none=True)
            torch.nn.functional.cross_entropy(m(x).float(), y).backward()
        for _ in range(3): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8): step()
        torch.cuda.synchronize(); res[mode] = (time.perf_counter() - t0) / 8
        del m; torch.cuda.empty_cache()
    print(f"{name}: bf16 {res['bf16'] * 1e3:.1f} ms/step ({B / res['bf16']:.0f} img/s) | fp8 forward operands {res['fp8-forward'] * 1e3:.1f} ms/step "
          f"({B / res['fp8-forward']:.0f} img/s, x{res['bf16'] / res['fp8-forward']:.3f})")
