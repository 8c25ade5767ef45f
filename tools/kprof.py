"""Run a few launches of selected kernels (for rocprofv3 --pmc): python tools/kprof.py"""
import torch
from vit_pytorch_amd import kernels as K, _lib as L
dev = "cuda"; BF = torch.bfloat16
M, D, F = 50432, 768, 3072
def run(n, k, epi, iters=3):
    A = torch.randn(M, k, device=dev).to(BF); W = (torch.randn(n, k, device=dev) * k ** -0.5).to(BF)
    bias = torch.randn(n, device=dev).to(BF)
    if epi == L.EPI_RESID:
        C = torch.zeros(M, n, device=dev); resid = C; aux = None
    else:
        C = torch.empty(M, n, dtype=BF, device=dev); resid = None; aux = torch.randn(M, n, device=dev).to(BF)
    for _ in range(iters):
        K.gemm_nt_bf16(A, k, W, k, C, n, M, n, k, epi, bias=bias, resid=resid, aux=aux)
    torch.cuda.synchronize()
run(3 * D, D, L.EPI_NONE); run(F, D, L.EPI_BIAS_GELU); run(D, D, L.EPI_RESID); run(D, F, L.EPI_RESID)
for n, k in ((3 * D, D), (F, D)):
    dY = torch.randn(M, n, device=dev).to(BF); X = torch.randn(M, k, device=dev).to(BF)
    s = K.gemm_tn_splits(M, n, k); ws = torch.empty(s * n * k, device=dev); dW = torch.empty(n, k, dtype=BF, device=dev)
    for _ in range(3):
        K.gemm_tn_bf16(dY, n, X, k, dW, k, M, n, k, ws, s)
torch.cuda.synchronize()
