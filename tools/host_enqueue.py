import time, torch
from vit_pytorch_amd import ViT
torch.manual_seed(0)
m = ViT(image_size=224, patch_size=16, num_classes=1000, dim=768, depth=12, heads=12, mlp_dim=3072).cuda().bfloat16()
import sys; B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = torch.randn(B, 3, 224, 224, device="cuda").bfloat16(); y = torch.randint(0, 1000, (B,), device="cuda")
def step():
    m.zero_grad(set_to_none=True)
    torch.nn.functional.cross_entropy(m(x).float(), y).backward()
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/20:.2f} ms/step; wall {1e3*(t2-t0)/20:.2f} ms/step")
