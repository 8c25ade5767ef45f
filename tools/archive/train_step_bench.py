"""Full training step (forward + backward + fused Adam) of ViT-B/16 at batch 256: what the weight-derived caches (K-blocked
copies, transposes) cost when the weights change every step.  python tools/train_step_bench.py [steps]
VITK_PACK_W=0 for the row-major weights."""
import sys, time, torch
from vit_pytorch_amd import ViT, optim, parallel

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = "cuda"
m = ViT(image_size=224, patch_size=16, num_classes=1000, dim=768, depth=12, heads=12, mlp_dim=3072).to(dev, dtype=torch.bfloat16)
dp = parallel.DataParallel(m)      # world size 1: the flat gradient buffer without collectives
opt = optim.Adam(dp, lr=1e-4)
img = torch.randn(256, 3, 224, 224, device=dev, dtype=torch.bfloat16)
lab = torch.randint(0, 1000, (256,), device=dev)


def step(with_opt=True):
    out = dp(img)
    loss = torch.nn.functional.cross_entropy(out.float(), lab)
    dp.backward(loss)
    if with_opt:
        opt.step()


for with_opt in (False, True, False, True):
    for _ in range(3):
        step(with_opt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        step(with_opt)
    torch.cuda.synchronize()
    print(f"{'fwd+bwd+adam' if with_opt else 'fwd+bwd     '}: {(time.perf_counter() - t0) / steps * 1e3:.2f} ms/step")
