"""Slot timing of the persistent NT kernel from s_memtime stamps (waves 0 and 4 of one workgroup, K-steps 8..39)."""
import os
import torch
from vit_pytorch_amd import kernels as K, _lib as L
dev = "cuda"; BF = torch.bfloat16
M = 50432
st = torch.zeros(2048, dtype=torch.int64, device=dev)
for name, (n, k) in {"qkv": (2304, 768), "dx ff1 (K=3072)": (768, 3072)}.items():
    A = torch.randn(M, k, device=dev).to(BF); W = (torch.randn(n, k, device=dev) * k ** -0.5).to(BF); C = torch.empty(M, n, dtype=BF, device=dev)
    for _ in range(2):
        K.gemm_nt_bf16(A, k, W, k, C, n, M, n, k)
    os.environ["VITK_NTP_STAMPS"] = str(st.data_ptr())
    K.gemm_nt_bf16(A, k, W, k, C, n, M, n, k)
    torch.cuda.synchronize()
    os.environ.pop("VITK_NTP_STAMPS")
    t = st.cpu().view(2, 128, 8)[:, :32]
    names = ["R0 reads issued", "R0 dma issued", "R0 lgkm waited", "barrier", "M0 issued", "barrier", "R1 (reads, dma, vmcnt, lgkm)", "barrier + M1 + barrier"]
    print(f"== {name}")
    for g in (0, 1):
        d = t[g]
        ok = (d[1:, 0] - d[:-1, 0]) < 20000           # skip the tile boundary (epilogue in between)
        seg = [((d[:, i + 1] - d[:, i]).float()[:-1][ok]).mean().item() for i in range(7)] + [((d[1:, 0] - d[:-1, 7]).float()[ok]).mean().item()]
        step = (d[1:, 0] - d[:-1, 0]).float()[ok].mean().item()
        print(f" group {'AB'[g]}: step {step:.0f} | " + " | ".join(f"{names[i]} {seg[i]:.0f}" for i in range(8)))
