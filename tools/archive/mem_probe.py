"""Step time, peak memory and allocator retries of one bench config:  python tools/mem_probe.py vit_h14 256 [steps]"""
import sys, time, torch
sys.path.insert(0, ".")
import bench
from vit_pytorch_amd import ViT, engine

name, batch = sys.argv[1], int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cfg = bench.CONFIGS[name][0]
m = ViT(**cfg).to("cuda", dtype=torch.bfloat16)
img = torch.randn(batch, 3, cfg["image_size"], cfg["image_size"], device="cuda", dtype=torch.bfloat16)
lab = torch.randint(0, 1000, (batch,), device="cuda")


def step():
    for p in m.parameters():
        p.grad = None
    torch.nn.functional.cross_entropy(m(img).float(), lab).backward()


step(); torch.cuda.synchronize()
torch.cuda.reset_peak_memory_stats()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
st = torch.cuda.memory_stats()
print(f"{name} batch {batch}: {ms:.1f} ms/step  peak allocated {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB  reserved {torch.cuda.max_memory_reserved() / 2**30:.1f} GiB"
      f"  alloc retries {st.get('num_alloc_retries')}  ooms {st.get('num_ooms')}  device total {torch.cuda.get_device_properties(0).total_memory / 2**30:.0f} GiB")
