"""LayerNorm forward on 16-bit rows at the transformer widths: python tools/ln16_ab.py   (VITK_LN_FWD16=0: the general kernel)"""
import os, torch
from vit_pytorch_amd import kernels as K
dev = "cuda"; BF = torch.bfloat16
for M, D in ((50432, 768), (25216, 1024), (147712, 1280)):
    x = (torch.randn(M, D, device=dev) * 2 + 0.3).to(BF); w = (1 + 0.1 * torch.randn(D, device=dev)).to(BF); b = (0.1 * torch.randn(D, device=dev)).to(BF)
    y = torch.empty(M, D, dtype=BF, device=dev); mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
    fn = lambda: K.layernorm_fwd(x, w, b, y, mean, rstd, M, D)
    for _ in range(5): fn()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20)
    t = sorted(ts)[2]
    print(f"[VITK_LN_FWD16={os.environ.get('VITK_LN_FWD16', '1')}] {M} x {D}: {t * 1e3:.1f} us = {M * D * 4 / t / 1e9:.2f} TB/s  checksum {y.double().sum().item():.6e}")
