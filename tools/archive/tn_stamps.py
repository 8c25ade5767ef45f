"""Slot timing of the TN DMA kernel from s_memtime stamps (waves 0 and 4 of one workgroup, K-steps 8..39)."""
import os
import torch
os.environ["VITK_TN_DMA"] = "1"
from vit_pytorch_amd import kernels as K
dev = "cuda"; BF = torch.bfloat16
M, n, k = 50432, 3072, 768
dY = torch.randn(M, n, device=dev).to(BF); X = torch.randn(M, k, device=dev).to(BF)
s = K.gemm_tn_splits(M, n, k); ws = torch.empty(s * n * k, device=dev); dW = torch.empty(n, k, dtype=BF, device=dev)
st = torch.zeros(2048, dtype=torch.int64, device=dev)
for dbg in (0, 2, 1):
    os.environ["VITK_TN_DBG"] = str(dbg)
    for _ in range(2):
        K.gemm_tn_bf16(dY, n, X, k, dW, k, M, n, k, ws, s)
    os.environ["VITK_TN_STAMPS"] = str(st.data_ptr())
    K.gemm_tn_bf16(dY, n, X, k, dW, k, M, n, k, ws, s)
    torch.cuda.synchronize()
    os.environ.pop("VITK_TN_STAMPS")
    t = st.cpu().view(2, 128, 8)[:, :32]          # [wave group][step][stamp]
    names = ["R0 reads issued", "R0 dma issued", "R0 lgkm waited", "barrier", "M0 issued", "barrier", "R1 (reads, dma, vmcnt, lgkm)", "barrier -> next step"]
    print(f"== dbg={dbg} (1 = no DMA in loop, 2 = no MFMA); cycles (100 MHz s_memtime ticks x?)")
    for g in (0, 1):
        d = t[g]
        seg = torch.zeros(8)
        for i in range(7):
            seg[i] = (d[:, i + 1] - d[:, i]).float().mean()
        seg[7] = (d[1:, 0] - d[:-1, 7]).float().mean()
        step = (d[1:, 0] - d[:-1, 0]).float().mean()
        print(f" group {'AB'[g]}: step {step:.0f} | " + " | ".join(f"{names[i]} {seg[i]:.0f}" for i in range(8)))
