"""A few launches of each GEMM of a ViT-B/16 batch-256 layer, in a FIXED ORDER, for rocprofv3 --pmc: python tools/kprof.py
tools/pmc_traffic_json.py labels the launch groups by this order (LABELS)."""
import os
import torch
from vit_pytorch_amd import kernels as K, _lib as L
dev = "cuda"; BF = torch.bfloat16
M, D, F = 50432, 768, 3072
ITERS = 3
# (label, N, K, algorithmic bytes per launch: operands once + output once (+ aux / residual once))
LABELS = []


def run(label, n, k, epi):
    A = torch.randn(M, k, device=dev).to(BF); Wrow = (torch.randn(n, k, device=dev) * k ** -0.5).to(BF)
    bias = torch.randn(n, device=dev).to(BF)
    alg = 2 * (M * k + n * k)
    # the operand production uses: the K-blocked copy (vitk_pack_w_nt; KPROF_ROW_MAJOR=1 for the plain weight)
    if os.environ.get("KPROF_ROW_MAJOR"):
        W, ldw = Wrow, k
    else:
        W, ldw = torch.empty(K.pack_w_nt_bytes(n, k) // 2, dtype=BF, device=dev), 0
        K.pack_w_nt(Wrow, k, n, k, W, None)
    if epi == L.EPI_RESID:
        C = torch.zeros(M, n, device=dev); resid = C; aux = None
        alg += 8 * M * n                                   # f32 residual read + f32 write
    elif epi == L.EPI_RESID16:
        C = torch.zeros(M, n, dtype=BF, device=dev); resid = C; aux = None
        alg += 4 * M * n                                   # 16-bit residual read + 16-bit write
    else:
        C = torch.empty(M, n, dtype=BF, device=dev); resid = None; aux = torch.randn(M, n, device=dev).to(BF)
        alg += 2 * M * n
    dg = not os.environ.get("KPROF_NO_DG")      # round 4 default: FF1 stores the gelu' factor (EPI_BIAS_GELU_DG), dFF1 multiplies (EPI_MUL_AUX)
    dg8 = dg and not os.environ.get("KPROF_DG16")       # round 5 default: the factor as 8-bit codes (EPI_BIAS_GELU_DG8 / EPI_MUL_AUX8)
    if epi in (L.EPI_BIAS_GELU, L.EPI_GELU_BWD):
        alg += (1 if dg8 else 2) * M * n           # + the factor / pre-activation written (FF1) / read (dFF1)
        if dg8:
            aux = torch.randint(1, 254, (M, n), dtype=torch.uint8, device=dev)
    for _ in range(ITERS):
        if epi == L.EPI_GELU_BWD:
            rows = K.gemm_nt_colsum_rows(M, n, k, n)
            cs = torch.empty(rows * n, device=dev)
            (K.gemm_nt_bf16_mul_aux8_colsum if dg8 else K.gemm_nt_bf16_mul_aux_colsum if dg else K.gemm_nt_bf16_gelu_bwd_colsum)(A, k, W, ldw, C, n, M, n, k, aux, cs)
        else:
            e = (L.EPI_BIAS_GELU_DG8 if dg8 else L.EPI_BIAS_GELU_DG) if (dg and epi == L.EPI_BIAS_GELU) else epi
            K.gemm_nt_bf16(A, k, W, ldw, C, n, M, n, k, e, bias=bias, resid=resid, aux=aux)
    torch.cuda.synchronize()
    LABELS.append((label, n, k, alg))


def run_tn(label, n, k):
    dY = torch.randn(M, n, device=dev).to(BF); X = torch.randn(M, k, device=dev).to(BF)
    s = K.gemm_tn_splits(M, n, k); ws = torch.empty(s * n * k, device=dev); dW = torch.empty(n, k, dtype=BF, device=dev)
    for _ in range(ITERS):
        K.gemm_tn_bf16(dY, n, X, k, dW, k, M, n, k, ws, s)
    torch.cuda.synchronize()
    LABELS.append((label, n, k, 2 * (M * n + M * k + n * k)))


# round 4: the forward residual stream is 16-bit by default (ops.fwd_stream_16), so the two residual GEMMs run the RESID16 epilogue;
# KPROF_F32_STREAM=1 profiles the float32-stream flavour (rounds 1-3)
_R = L.EPI_RESID if os.environ.get("KPROF_F32_STREAM") else L.EPI_RESID16
_RN = "f32 residual" if os.environ.get("KPROF_F32_STREAM") else "16-bit residual"
PLAN = (("QKV", 3 * D, D, L.EPI_NONE), ("FF1 bias+GELU", F, D, L.EPI_BIAS_GELU), (f"out-proj + {_RN}", D, D, _R),
        (f"FF2 + {_RN}", D, F, _R), ("dFF1 GELU' + column sums", F, D, L.EPI_GELU_BWD), ("dX of FF1 (K=3072)", D, F, L.EPI_NONE))
TN_PLAN = (("dW qkv", 3 * D, D), ("dW ff1", F, D))

if __name__ == "__main__":
    for a in PLAN:
        run(*a)
    for a in TN_PLAN:
        run_tn(*a)
    torch.cuda.synchronize()
