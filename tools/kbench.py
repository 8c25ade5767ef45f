"""Kernel micro-benchmarks at BASELINE config-2 sizes (ViT-B/16, B=256): python tools/kbench.py [B]"""
import sys
import torch
from vit_pytorch_amd import kernels as K, _lib as L

dev = "cuda"
BF = torch.bfloat16


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters  # ms


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 256
    N, D, F, H, d = 197, 768, 3072, 12, 64
    M = B * N
    print(f"device={torch.cuda.get_device_name(0)} B={B} M={M}")
    # ---- NT GEMMs
    for name, (n, k, epi) in {"qkv": (3 * D, D, L.EPI_NONE), "out+resid": (D, D, L.EPI_RESID), "ff1+gelu": (F, D, L.EPI_BIAS_GELU),
                              "ff2+resid": (D, F, L.EPI_RESID), "dff1(gelu_bwd)": (F, D, L.EPI_GELU_BWD)}.items():
        A = torch.randn(M, k, device=dev).to(BF); W = (torch.randn(n, k, device=dev) * k ** -0.5).to(BF)
        bias = torch.randn(n, device=dev).to(BF)
        if epi == L.EPI_RESID:
            C = torch.zeros(M, n, device=dev); resid = C; aux = None
        else:
            C = torch.empty(M, n, dtype=BF, device=dev); resid = None; aux = torch.randn(M, n, device=dev).to(BF)
        t = timeit(lambda: K.gemm_nt_bf16(A, k, W, k, C, n, M, n, k, epi, bias=bias, resid=resid, aux=aux))
        print(f"gemm_nt {name:16s} M={M} N={n} K={k}: {t:.3f} ms  {2 * M * n * k / t / 1e9:.1f} TF/s")
        t = timeit(lambda: torch.matmul(A, W.t()))
        print(f"   torch(hipBLASLt) plain        : {t:.3f} ms  {2 * M * n * k / t / 1e9:.1f} TF/s")
    # ---- TN GEMMs
    for name, (n, k) in {"dWqkv": (3 * D, D), "dWout": (D, D), "dW1": (F, D), "dW2": (D, F)}.items():
        dY = torch.randn(M, n, device=dev).to(BF); X = torch.randn(M, k, device=dev).to(BF)
        splits = K.gemm_tn_splits(M, n, k)
        ws = torch.empty(splits * n * k, device=dev); dW = torch.empty(n, k, dtype=BF, device=dev)
        t = timeit(lambda: K.gemm_tn_bf16(dY, n, X, k, dW, k, M, n, k, ws, splits))
        print(f"gemm_tn {name:8s} N={n} K={k} splits={splits}: {t:.3f} ms  {2 * M * n * k / t / 1e9:.1f} TF/s")
        t = timeit(lambda: torch.matmul(dY.t(), X))
        print(f"   torch(hipBLASLt)              : {t:.3f} ms  {2 * M * n * k / t / 1e9:.1f} TF/s")
    # ---- attention
    I = H * d
    qkv = torch.randn(B, N, 3 * I, device=dev).to(BF); o = torch.empty(B, N, I, dtype=BF, device=dev)
    lse = torch.empty(B, H, N, device=dev); delta = torch.empty(B, H, N, device=dev)
    do = torch.randn(B, N, I, device=dev).to(BF); dqkv = torch.empty_like(qkv)
    sb, sh, sn = N * 3 * I, d, 3 * I
    q_ = K.bhnd(qkv, sb, sh, sn); k_ = K.bhnd(qkv, sb, sh, sn, offset=I); v_ = K.bhnd(qkv, sb, sh, sn, offset=2 * I)
    o_ = K.bhnd(o, N * I, d, I)
    t = timeit(lambda: K.attn_fwd_bf16(q_, k_, v_, o_, lse, B, H, N, d, d ** -0.5))
    fl = 4 * B * H * N * N * d
    print(f"attn_fwd: {t:.3f} ms  {fl / t / 1e9:.1f} TF/s (algorithmic)  {(4 * B * N * I * 2) / t / 1e6:.0f} GB/s")
    t = timeit(lambda: K.attn_bwd_bf16(q_, k_, v_, o_, K.bhnd(do, N * I, d, I), lse, delta, K.bhnd(dqkv, sb, sh, sn),
                                       K.bhnd(dqkv, sb, sh, sn, offset=I), K.bhnd(dqkv, sb, sh, sn, offset=2 * I), B, H, N, d, d ** -0.5))
    print(f"attn_bwd: {t:.3f} ms  {2.5 * fl / t / 1e9:.1f} TF/s (algorithmic 2.5x fwd)")
    qh = qkv[..., :I].reshape(B, N, H, d).transpose(1, 2); kh = qkv[..., I:2 * I].reshape(B, N, H, d).transpose(1, 2); vh = qkv[..., 2 * I:].reshape(B, N, H, d).transpose(1, 2)
    try:
        t = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(qh, kh, vh))
        print(f"   torch sdpa fwd: {t:.3f} ms")
    except Exception as e:  # noqa
        print("   torch sdpa failed:", e)
    # ---- LayerNorm
    x = torch.randn(M, D, device=dev); w = torch.ones(D, dtype=BF, device=dev); b = torch.zeros(D, dtype=BF, device=dev)
    y = torch.empty(M, D, dtype=BF, device=dev); mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
    t = timeit(lambda: K.layernorm_fwd(x, w, b, y, mean, rstd, M, D))
    print(f"layernorm_fwd f32->bf16: {t:.3f} ms  {M * D * 6 / t / 1e6:.0f} GB/s")
    nblk = K.layernorm_bwd_blocks(M, D); partials = torch.empty(3 * nblk * D, device=dev)
    dxf = torch.empty(M, D, device=dev); dxt = torch.empty(M, D, dtype=BF, device=dev); gin = torch.randn(M, D, device=dev)
    t = timeit(lambda: K.layernorm_bwd(y, x, w, mean, rstd, gin, dxf, dxt, partials, True, M, D))
    print(f"layernorm_bwd: {t:.3f} ms  {M * D * (2 + 4 + 4 + 4 + 2) / t / 1e6:.0f} GB/s")
    ws = torch.empty(K.colsum_ws_floats(M, F), device=dev); hbig = torch.randn(M, F, device=dev).to(BF); outc = torch.empty(F, dtype=BF, device=dev)
    t = timeit(lambda: K.colsum(hbig, M, F, F, outc, ws))
    print(f"colsum M x 3072 bf16: {t:.3f} ms  {M * F * 2 / t / 1e6:.0f} GB/s")


if __name__ == "__main__" and not any(a.startswith("--") for a in sys.argv[1:]):
    main()


def ln_wide():
    """LayerNorm backward at the ViT-L / NaViT row width (D = 1024: the 4-wave block variant)."""
    M, D = 25216, 1024
    x = torch.randn(M, D, device=dev); w = torch.ones(D, dtype=BF, device=dev); y = torch.randn(M, D, device=dev).to(BF)
    mean = torch.zeros(M, device=dev); rstd = torch.ones(M, device=dev); gin = torch.randn(M, D, device=dev)
    nblk = K.layernorm_bwd_blocks(M, D); partials = torch.empty(3 * nblk * D, device=dev)
    dxf = torch.empty(M, D, device=dev); dxt = torch.empty(M, D, dtype=BF, device=dev)
    t = timeit(lambda: K.layernorm_bwd(y, x, w, mean, rstd, gin, dxf, dxt, partials, True, M, D))
    print(f"layernorm_bwd D=1024 M={M}: {t:.3f} ms  {M * D * 16 / t / 1e6:.0f} GB/s")


if __name__ == "__main__" and "--ln-wide" in sys.argv:
    ln_wide()


def fp8():
    """fp8 (e4m3) operand NT GEMM against the bf16 one at the config-2 and config-5 FeedForward shapes."""
    for name, (M, n, k) in {"vit-b ff1": (50432, 3072, 768), "vit-b ff2": (50432, 768, 3072), "vit-b qkv": (50432, 2304, 768),
                            "vit-h ff1 (b64)": (36928, 5120, 1280), "vit-h ff2 (b64)": (36928, 1280, 5120)}.items():
        A = torch.randn(M, k, device=dev).to(BF); W = (torch.randn(n, k, device=dev) * k ** -0.5).to(BF)
        C = torch.empty(M, n, dtype=BF, device=dev)
        A8 = torch.empty(M, k, dtype=torch.uint8, device=dev); W8 = torch.empty(n, k, dtype=torch.uint8, device=dev)
        sa = torch.empty(2, device=dev); sw = torch.empty(2, device=dev)
        K.fp8_amax_scale(A, sa); K.fp8_amax_scale(W, sw); K.quantize_fp8(A, A8, scale_dev=sa); K.quantize_fp8(W, W8, scale_dev=sw)
        t16 = timeit(lambda: K.gemm_nt_bf16(A, k, W, k, C, n, M, n, k))
        t8 = timeit(lambda: K.gemm_nt_fp8(A8, k, W8, k, C, n, M, n, k, 1.0))
        tq = timeit(lambda: (K.fp8_amax_scale(A, sa), K.quantize_fp8(A, A8, scale_dev=sa)))
        fl = 2 * M * n * k
        print(f"{name:16s} M={M} N={n} K={k}: bf16 {t16:.3f} ms {fl / t16 / 1e9:6.0f} TF/s | fp8 {t8:.3f} ms {fl / t8 / 1e9:6.0f} TF/s"
              f" (x{t16 / t8:.2f}) | amax+quantise A {tq:.3f} ms")


if __name__ == "__main__" and "--fp8" in sys.argv:
    fp8()
