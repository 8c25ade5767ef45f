"""A few launches of both TN kernels (for rocprofv3 --pmc): python tools/kprof_tn.py"""
import os
import torch
from vit_pytorch_amd import kernels as K
dev = "cuda"; BF = torch.bfloat16
M, D, F = 50432, 768, 3072
for n, k in ((3 * D, D), (F, D)):
    dY = torch.randn(M, n, device=dev).to(BF); X = torch.randn(M, k, device=dev).to(BF)
    s = K.gemm_tn_splits(M, n, k); ws = torch.empty(s * n * k, device=dev); dW = torch.empty(n, k, dtype=BF, device=dev)
    for old in (0, 1):
        if old:
            os.environ.pop("VITK_TN_DMA", None)
        else:
            os.environ["VITK_TN_DMA"] = "1"
        for _ in range(3):
            K.gemm_tn_bf16(dY, n, X, k, dW, k, M, n, k, ws, s)
torch.cuda.synchronize()
