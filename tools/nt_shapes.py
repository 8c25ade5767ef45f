"""The eight NT GEMMs of a ViT-B/16 layer (batch 256, K-blocked weights) on the loaded library: python tools/nt_shapes.py [rounds]
(VITK_LIB=<other build> for an A/B across builds: run the two alternately in one gpurun call)."""
import os
import statistics
import sys
import torch
os.environ.setdefault("VITK_NTP_EPIS", "31")
from vit_pytorch_amd import kernels as K, _lib as L

dev = "cuda"; BF = torch.bfloat16


def time_once(fn, iters=10):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B, N, D, F = 256, 197, 768, 3072
M = B * N
shapes = {"qkv": (3 * D, D, L.EPI_NONE), "out+resid": (D, D, L.EPI_RESID), "ff1+gelu": (F, D, L.EPI_BIAS_GELU), "ff2+resid": (D, F, L.EPI_RESID),
          "dff1": (F, D, L.EPI_GELU_BWD), "dx_ff1": (D, F, L.EPI_NONE), "dx_qkv": (D, 3 * D, L.EPI_NONE), "dx_out": (D, D, L.EPI_NONE)}
out = []
tot = 0.0
for name, (n, k, epi) in shapes.items():
    A = torch.randn(M, k, device=dev).to(BF); W = (torch.randn(n, k, device=dev) * k ** -0.5).to(BF)
    bias = torch.randn(n, device=dev).to(BF)
    Wp = torch.empty(K.pack_w_nt_bytes(n, k) // 2, dtype=BF, device=dev)
    K.pack_w_nt(W, k, n, k, Wp, None)
    if epi == L.EPI_RESID:
        C = torch.zeros(M, n, device=dev); resid = C; aux = None
    else:
        C = torch.empty(M, n, dtype=BF, device=dev); resid = None; aux = torch.randn(M, n, device=dev).to(BF)
    part = torch.empty(K.gemm_nt_colsum_rows(M, n, k, n) * n, device=dev) if epi == L.EPI_GELU_BWD else None

    def run():
        if epi == L.EPI_GELU_BWD:
            K.gemm_nt_bf16_gelu_bwd_colsum(A, k, Wp, 0, C, n, M, n, k, aux, part)
        else:
            K.gemm_nt_bf16(A, k, Wp, 0, C, n, M, n, k, epi, bias=bias if epi in (L.EPI_BIAS, L.EPI_BIAS_GELU, L.EPI_RESID) else None, resid=resid, aux=aux)
    ts = [time_once(run) for _ in range(rounds + 1)][1:]
    t = statistics.median(ts)
    tot += t
    out.append(f"{name} {t:.1f}")
print(f"[{os.path.basename(os.environ.get('VITK_LIB', 'libvitk.so'))}] " + " | ".join(out) + f" | sum {tot:.1f} us")
