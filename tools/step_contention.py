"""The whole ViT-B/16 training step (batch 256, fwd + bwd) while another resident kernel holds c CUs for the entire step -- a stand-in
for an RCCL collective (vitk_test_occupy_cus on a priority -1 stream; one GPU).  Modes: the default single-GPU configuration
(static tile lists, weight-gradient GEMMs planned for 256 CUs) vs what parallel.FlatGradSink switches on around a collective
(vitk_set_cu_reserve(c): tile tickets in the persistent NT GEMMs, weight-gradient GEMMs planned for 256 - c CUs, per-head attention
kernels).  Ideal slowdown: 256 / (256 - c).      python tools/step_contention.py"""
import statistics
import torch
from vit_pytorch_amd import ViT, _lib as L

dev = "cuda"
lib = L.load()
torch.manual_seed(0)
m = ViT(image_size=224, patch_size=16, num_classes=1000, dim=768, depth=12, heads=12, mlp_dim=3072).to(dev, dtype=torch.bfloat16)
img = torch.randn(256, 3, 224, 224, device=dev).to(torch.bfloat16)
lab = torch.randint(0, 1000, (256,), device=dev)
hog = torch.cuda.Stream(priority=-1)


def step():
    for p in m.parameters():
        p.grad = None
    torch.nn.functional.cross_entropy(m(img).float(), lab).backward()


def timed(c, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        if c:
            L.check(lib.vitk_test_occupy_cus(c, 80.0, hog.cuda_stream), "occupy")      # outlives the step
            torch.cuda._sleep(300000)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); step(); e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
        torch.cuda.synchronize()
    return statistics.median(ts)


for _ in range(3):
    step()
base = timed(0)
print(f"device={torch.cuda.get_device_name(0)}  step with no other kernel: {base:.2f} ms")
for c in (16, 32):
    L.check(lib.vitk_set_cu_reserve(0), "reserve")
    t_plain = timed(c)
    L.check(lib.vitk_set_cu_reserve(c), "reserve")
    t_res = timed(c)
    L.check(lib.vitk_set_cu_reserve(0), "reserve")
    print(f"  {c} CUs held for the whole step: default configuration {t_plain:.2f} ms (x{t_plain / base:.3f})   with vitk_set_cu_reserve({c}) "
          f"{t_res:.2f} ms (x{t_res / base:.3f})   ideal x{256 / (256 - c):.3f} ({c / 256 * 100:.1f} % of the CUs)")
