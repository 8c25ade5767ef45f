"""ViT-B/16 224^2 batch 256 fwd+bwd: bfloat16 model vs float32 model under torch.autocast(bfloat16) (functional.autocast_aware: the 16-bit
engine on per-forward 16-bit parameter copies), same box, interleaved windows.  python tools/autocast_ab.py [steps]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from vit_pytorch_amd import ViT

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cfg = dict(image_size=224, patch_size=16, num_classes=1000, dim=768, depth=12, heads=12, mlp_dim=3072)
torch.manual_seed(0)
m16 = ViT(**cfg).to("cuda", dtype=torch.bfloat16)
m32 = ViT(**cfg).to("cuda")
x16 = torch.randn(256, 3, 224, 224, device="cuda").to(torch.bfloat16); x32 = x16.float()
y = torch.randint(0, 1000, (256,), device="cuda")
lossf = torch.nn.functional.cross_entropy


def step16():
    m16.zero_grad(set_to_none=True); lossf(m16(x16).float(), y).backward()


def step_ac():
    m32.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = m32(x32)
    lossf(out.float(), y).backward()


def timeit(fn):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps * 1e3


for f in (step16, step_ac):
    for _ in range(3):
        f()
res = {"bf16": [], "autocast": [], "autocast, VITK_AUTOCAST_STREAM=16 (round 5: streams in the parameter dtype)": []}
for _ in range(3):
    res["bf16"].append(round(timeit(step16), 3)); res["autocast"].append(round(timeit(step_ac), 3))
    os.environ["VITK_AUTOCAST_STREAM"] = "16"
    step_ac()
    res["autocast, VITK_AUTOCAST_STREAM=16 (round 5: streams in the parameter dtype)"].append(round(timeit(step_ac), 3))
    del os.environ["VITK_AUTOCAST_STREAM"]
    step_ac()
print("ms/step", res, "peak GiB", round(torch.cuda.max_memory_allocated() / 2 ** 30, 1))
