"""NaViT (BASELINE config 4) timing: packed variable-resolution batch, dim 1024 / depth 24 / heads 16 / mlp 4096,
images a*16 x b*16 px with a, b ~ U{4..40}, ~32k tokens.  python tools/navit_bench.py [depth] [target_tokens]"""
import sys, time
import numpy as np
import torch
from vit_pytorch_amd.na_vit import NaViT

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 24
target = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
dev = "cuda"
rng = np.random.default_rng(0)
sizes = []
tok = 0
while tok < target:
    a, b = rng.integers(4, 41, 2)
    sizes.append((int(a) * 16, int(b) * 16)); tok += int(a) * int(b)
torch.manual_seed(0)
m = NaViT(image_size=1024, patch_size=16, num_classes=1000, dim=1024, depth=depth, heads=16, mlp_dim=4096).to(dev, dtype=torch.bfloat16)
imgs = [torch.randn(3, h, w, device=dev).to(torch.bfloat16) for h, w in sizes]
labels = torch.randint(0, 1000, (len(imgs),), device=dev)

def step():
    m.zero_grad(set_to_none=True)
    out = m(imgs, group_images=True, group_max_seq_len=4096)
    torch.nn.functional.cross_entropy(out.float(), labels).backward()

for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 5
for _ in range(n): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
D, F, I = 1024, 4096, 1024
lens = [(h // 16) * (w // 16) for h, w in sizes]
gemm = depth * tok * (2 * D * 3 * I + 2 * I * D + 4 * D * F)
attn = depth * sum(4 * 16 * n_ * n_ * 64 for n_ in lens)
print(f"NaViT cfg4: {len(imgs)} images, {tok} tokens (max {max(lens)}/image), depth {depth}: {dt*1e3:.1f} ms/step, "
      f"{tok/dt:.0f} tokens/s, {len(imgs)/dt:.1f} img/s, {3*(gemm+attn)/dt/1e12:.1f} TFLOP/s algorithmic "
      f"(attention share of FLOPs {attn/(gemm+attn)*100:.1f}%)")
