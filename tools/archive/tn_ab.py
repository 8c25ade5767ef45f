"""A/B of the weight-gradient (TN) GEMM kernels at the ViT-B/16 (batch 256) shapes: LDS-DMA ping-pong kernel (gemm_tn_dma.hip) vs
the register-staged default, interleaved rounds in one process; times include the slab reduction.
    python tools/tn_ab.py [rounds]"""
import os
import statistics
import sys
import torch
from vit_pytorch_amd import kernels as K

dev = "cuda"; BF = torch.bfloat16


def time_once(fn, iters=10):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    M, D, F = 256 * 197, 768, 3072
    tot_n = tot_o = 0.0
    for name, (n, k) in {"dWqkv": (3 * D, D), "dWout": (D, D), "dW1": (F, D), "dW2": (D, F)}.items():
        dY = torch.randn(M, n, device=dev).to(BF); X = torch.randn(M, k, device=dev).to(BF)
        splits = K.gemm_tn_splits(M, n, k)
        ws = torch.empty(splits * n * k, device=dev); dW = torch.empty(n, k, dtype=BF, device=dev)
        run = lambda: K.gemm_tn_bf16(dY, n, X, k, dW, k, M, n, k, ws, splits)
        tn, to = [], []
        for r in range(rounds + 1):
            os.environ["VITK_TN_DMA"] = "1"; a = time_once(run)
            os.environ.pop("VITK_TN_DMA", None); b = time_once(run)
            if r:
                tn.append(a); to.append(b)
        a, b = statistics.median(tn), statistics.median(to)
        fl = 2 * M * n * k
        tot_n += a; tot_o += b
        print(f"{name:6s} N={n:5d} K={k:5d} splits={splits:3d}: dma {a * 1e3:7.1f} us {fl / a / 1e9:7.1f} TF/s | register-staged {b * 1e3:7.1f} us {fl / b / 1e9:7.1f} TF/s | x{b / a:.3f}")
    print(f"sum: dma {tot_n:.3f} ms, register-staged {tot_o:.3f} ms (x{tot_o / tot_n:.3f}); x12 layers = {12 * tot_n:.2f} vs {12 * tot_o:.2f} ms")


if __name__ == "__main__":
    main()
