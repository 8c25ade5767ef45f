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
# round 4 epilogues (16-bit residual stream, FF1 stores the gelu' factor, dFF1 multiplies by it); NT_SHAPES_R3=1: the round-3 set
R3 = bool(os.environ.get("NT_SHAPES_R3"))
RES, GELU = (L.EPI_RESID, L.EPI_BIAS_GELU) if R3 else (L.EPI_RESID16, L.EPI_BIAS_GELU_DG)
shapes = {"qkv": (3 * D, D, L.EPI_NONE), "out+resid": (D, D, RES), "ff1+gelu": (F, D, GELU), "ff2+resid": (D, F, RES),
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
    elif epi == L.EPI_RESID16:
        C = torch.zeros(M, n, dtype=BF, device=dev); resid = C; aux = None
    else:
        C = torch.empty(M, n, dtype=BF, device=dev); resid = None; aux = torch.randn(M, n, device=dev).to(BF)
    part = torch.empty(K.gemm_nt_colsum_rows(M, n, k, n) * n, device=dev) if epi == L.EPI_GELU_BWD else None

    def run():
        if epi == L.EPI_GELU_BWD:
            (K.gemm_nt_bf16_gelu_bwd_colsum if R3 else K.gemm_nt_bf16_mul_aux_colsum)(A, k, Wp, 0, C, n, M, n, k, aux, part)
        else:
            K.gemm_nt_bf16(A, k, Wp, 0, C, n, M, n, k, epi, bias=bias if epi in (L.EPI_BIAS, L.EPI_BIAS_GELU, L.EPI_BIAS_GELU_DG, L.EPI_RESID, L.EPI_RESID16) else None, resid=resid, aux=aux)
    ts = [time_once(run) for _ in range(rounds + 1)][1:]
    t = statistics.median(ts)
    tot += t
    out.append(f"{name} {t:.1f}")
tag = " ".join(f"{k[9:]}={v}" for k, v in sorted(os.environ.items()) if k.startswith("VITK_NTP_") and k != "VITK_NTP_EPIS")
print(f"[{os.path.basename(os.environ.get('VITK_LIB', 'libvitk.so'))} {tag}] " + " | ".join(out) + f" | sum {tot:.1f} us")
