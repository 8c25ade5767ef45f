"""LayerNorm forward / backward at the ViT-B/16 batch-256 shape under the variant switches of layernorm.hip
(VITK_LNB_VAR, VITK_LNB_BLOCKS, VITK_LNF_NT, VITK_LNF_BLOCKS: read once per process):  python tools/ln_ab.py
Prints the time per launch and checksums of every output (they must not depend on the variant)."""
import os, torch
from vit_pytorch_amd import kernels as K
dev = "cuda"; BF = torch.bfloat16
M, D = 50432, 768


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


torch.manual_seed(0)
x = torch.randn(M, D, device=dev) * 2 + 0.3; w = (1 + 0.1 * torch.randn(D, device=dev)).to(BF); b = (0.1 * torch.randn(D, device=dev)).to(BF)
y = torch.empty(M, D, dtype=BF, device=dev); mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
tf = timeit(lambda: K.layernorm_fwd(x, w, b, y, mean, rstd, M, D))
dy = torch.randn(M, D, device=dev).to(BF); gin = torch.randn(M, D, device=dev)
nblk = K.layernorm_bwd_blocks(M, D); partials = torch.zeros(3 * nblk * D, device=dev)
dxf = torch.empty(M, D, device=dev); dxt = torch.empty(M, D, dtype=BF, device=dev)
mode = os.environ.get("LNAB_MODE", "")
if mode == "inplace":
    dxf = gin                      # the stream gradient is updated in place
elif mode == "nogin":
    gin = None
elif mode == "nodxt":
    dxt = None
elif mode == "nodxf":
    dxf = None
elif mode == "xbf16":
    x = x.to(BF)
tb = timeit(lambda: K.layernorm_bwd(dy, x, w, mean, rstd, gin, dxf, dxt, partials, dxt is not None, M, D))
if dxf is None: dxf = torch.zeros(1, device=dev)
if dxt is None: dxt = torch.zeros(1, device=dev)
cs = partials.view(3, nblk, D).double().sum(1)
tag = " ".join(f"{k}={os.environ[k]}" for k in ("LNAB_MODE", "VITK_LNB_FAST", "VITK_LNB_NT", "VITK_LNB_PIPE", "VITK_LNB_VAR", "VITK_LNB_BLOCKS", "VITK_LNF_NT", "VITK_LNF_BLOCKS") if k in os.environ) or "default"
print(f"[{tag}] fwd {tf * 1e3:.1f} us ({M * D * 6 / tf / 1e9:.2f} TB/s)  bwd {tb * 1e3:.1f} us ({M * D * 16 / tb / 1e9:.2f} TB/s)  blocks {nblk}  "
      f"y {y.double().sum().item():.6e} dxf {dxf.double().abs().sum().item():.8e} dxt {dxt.double().abs().sum().item():.8e} "
      f"dgamma {cs[0].abs().sum().item():.8e} dbeta {cs[1].abs().sum().item():.8e} colsum {cs[2].abs().sum().item():.8e}")
