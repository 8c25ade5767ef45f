"""Co-scheduling on ONE GPU: how the big GEMMs behave while another resident kernel holds c CUs (a stand-in for an RCCL collective
of the data-parallel step: vitk_test_occupy_cus on a high-priority stream).

    python tools/cu_contention.py            -> per launch: the eight NT GEMMs of a ViT-B/16 layer with static tile lists vs dynamic
                                                tickets, the four weight-gradient GEMMs with and without vitk_set_cu_reserve(c)
Ideal slowdown with c CUs gone: 256 / (256 - c).  A static full-chip launch pays up to 2x (its late workgroups are a second round)."""
import os
import statistics
import torch
from vit_pytorch_amd import kernels as K, _lib as L

dev = "cuda"; BF = torch.bfloat16
B, N, D, F = 256, 197, 768, 3072
M = B * N
lib = L.load()
hog_stream = torch.cuda.Stream(priority=-1)


def timed(fn, c, reps=6):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        if c:
            L.check(lib.vitk_test_occupy_cus(c, 8.0, hog_stream.cuda_stream), "occupy")
            torch.cuda._sleep(200000)           # let the hog become resident first
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)


nt_shapes = {"qkv": (3 * D, D, L.EPI_NONE), "out+resid": (D, D, L.EPI_RESID), "ff1+gelu": (F, D, L.EPI_BIAS_GELU), "ff2+resid": (D, F, L.EPI_RESID),
             "dff1": (F, D, L.EPI_GELU_BWD), "dx_ff1": (D, F, L.EPI_NONE), "dx_qkv": (D, 3 * D, L.EPI_NONE), "dx_out": (D, D, L.EPI_NONE)}
runs = {}
for name, (n, k, epi) in nt_shapes.items():
    A = torch.randn(M, k, device=dev).to(BF); W = (torch.randn(n, k, device=dev) * k ** -0.5).to(BF); bias = torch.randn(n, device=dev).to(BF)
    Wp = torch.empty(K.pack_w_nt_bytes(n, k) // 2, dtype=BF, device=dev); K.pack_w_nt(W, k, n, k, Wp, None)
    if epi == L.EPI_RESID:
        C = torch.zeros(M, n, device=dev); resid = C; aux = None
    else:
        C = torch.empty(M, n, dtype=BF, device=dev); resid = None; aux = torch.randn(M, n, device=dev).to(BF)
    part = torch.empty(K.gemm_nt_colsum_rows(M, n, k, n) * n, device=dev) if epi == L.EPI_GELU_BWD else None

    def run(A=A, Wp=Wp, C=C, n=n, k=k, epi=epi, bias=bias, resid=resid, aux=aux, part=part):
        if epi == L.EPI_GELU_BWD:
            K.gemm_nt_bf16_gelu_bwd_colsum(A, k, Wp, 0, C, n, M, n, k, aux, part)
        else:
            K.gemm_nt_bf16(A, k, Wp, 0, C, n, M, n, k, epi, bias=bias if epi in (L.EPI_BIAS, L.EPI_BIAS_GELU, L.EPI_RESID) else None, resid=resid, aux=aux)
    runs[name] = run


def layer_nt():
    for r in runs.values():
        r()


print(f"device={torch.cuda.get_device_name(0)}  eight NT GEMMs of a ViT-B/16 layer (batch 256), us per layer")
for c in (0, 16, 32, 64):
    os.environ["VITK_NTP_STATIC"] = "1"; t_s = timed(layer_nt, c)
    os.environ.pop("VITK_NTP_STATIC"); os.environ["VITK_NTP_DYNAMIC"] = "1"; t_d = timed(layer_nt, c)
    os.environ.pop("VITK_NTP_DYNAMIC")
    print(f"  {c:3d} CUs held: static lists {t_s:8.1f}   dynamic tickets {t_d:8.1f}   (ideal x{256 / (256 - c):.3f})")

tn_shapes = {"dWqkv": (3 * D, D), "dWout": (D, D), "dW1": (F, D), "dW2": (D, F)}
tn = {}
for name, (n, k) in tn_shapes.items():
    tn[name] = (torch.randn(M, n, device=dev).to(BF), torch.randn(M, k, device=dev).to(BF), torch.empty(n, k, dtype=BF, device=dev), n, k)


def layer_tn():
    for dY, X, dW, n, k in tn.values():
        splits = K.gemm_tn_splits(M, n, k)
        ws = torch.empty(splits * n * k, device=dev)
        K.gemm_tn_bf16(dY, n, X, k, dW, k, M, n, k, ws, splits)


print("four weight-gradient GEMMs of a layer, us per layer")
for c in (0, 16, 32, 64):
    L.check(lib.vitk_set_cu_reserve(0), "reserve"); t0 = timed(layer_tn, c)
    L.check(lib.vitk_set_cu_reserve(c), "reserve"); t1 = timed(layer_tn, c)
    print(f"  {c:3d} CUs held: planned for 256 CUs {t0:8.1f}   planned for {256 - c} CUs {t1:8.1f}   (ideal x{256 / (256 - c):.3f})")
L.check(lib.vitk_set_cu_reserve(0), "reserve")
