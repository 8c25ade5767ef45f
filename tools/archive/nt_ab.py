"""A/B of the NT GEMM at the ViT-B/16 (batch 256) shapes of one layer's forward and backward, interleaved rounds in one process:
persistent kernel reading the K-blocked weight copy (vitk_pack_w_nt; production) vs the same kernel on the row-major weight
(NTAB_VS=rowmajor, default) or vs the per-tile kernel (NTAB_VS=pertile: VITK_NO_PERSIST=1, row-major both).
    python tools/nt_ab.py [rounds]
Process-level switches (read once): VITK_NTP_TAIL=-1 (no 128-row tail tiles), VITK_NTP_EFULL / VITK_NTP_EHALF (plan cost model),
VITK_NTP_DBG=1 (main loop alone)."""
import os
import sys
import statistics
import torch
os.environ.setdefault("VITK_NTP_EPIS", "31")
from vit_pytorch_amd import kernels as K, _lib as L

dev = "cuda"
BF = torch.bfloat16


def time_once(fn, iters=10):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    B, N, D, F = 256, 197, 768, 3072
    M = B * N
    shapes = {
        "qkv            ": (3 * D, D, L.EPI_NONE),
        "out+resid      ": (D, D, L.EPI_RESID),
        "ff1+bias+gelu  ": (F, D, L.EPI_BIAS_GELU),
        "ff2+resid      ": (D, F, L.EPI_RESID),
        "dff1 gelu'+csum": (F, D, L.EPI_GELU_BWD),
        "dx ff1 (K=3072)": (D, F, L.EPI_NONE),
        "dx qkv (K=2304)": (D, 3 * D, L.EPI_NONE),
        "dx out (K=768) ": (D, D, L.EPI_NONE),
    }
    print(f"device={torch.cuda.get_device_name(0)} M={M} noexact={os.environ.get('VITK_NTP_NOEXACT')} tail={os.environ.get('VITK_NTP_TAIL')}")
    tot_new = tot_old = 0.0
    for name, (n, k, epi) in shapes.items():
        A = torch.randn(M, k, device=dev).to(BF); W = (torch.randn(n, k, device=dev) * k ** -0.5).to(BF)
        bias = torch.randn(n, device=dev).to(BF)
        part = None
        if epi == L.EPI_RESID:
            C = torch.zeros(M, n, device=dev); resid = C; aux = None
        else:
            C = torch.empty(M, n, dtype=BF, device=dev); resid = None; aux = torch.randn(M, n, device=dev).to(BF)

        vs_pertile = os.environ.get("NTAB_VS", "rowmajor") == "pertile"
        Wp = torch.empty(K.pack_w_nt_bytes(n, k) // 2, dtype=BF, device=dev)
        K.pack_w_nt(W, k, n, k, Wp, None)
        op = [Wp, 0]

        def run():
            W, ldw = op
            if epi == L.EPI_GELU_BWD:
                R = K.gemm_nt_colsum_rows(M, n, k, n)
                nonlocal part
                if part is None or part.numel() != R * n:
                    part = torch.empty(R * n, device=dev)
                K.gemm_nt_bf16_gelu_bwd_colsum(A, k, W, ldw, C, n, M, n, k, aux, part)
            else:
                K.gemm_nt_bf16(A, k, W, ldw, C, n, M, n, k, epi, bias=bias if epi in (L.EPI_BIAS, L.EPI_BIAS_GELU, L.EPI_RESID) else None,
                               resid=resid, aux=aux)
        tn, to = [], []
        for r in range(rounds + 1):
            op[:] = [W if vs_pertile else Wp, k if vs_pertile else 0]
            t1 = time_once(run)
            op[:] = [W, k]
            if vs_pertile:
                os.environ["VITK_NO_PERSIST"] = "1"
            t0 = time_once(run)
            os.environ.pop("VITK_NO_PERSIST", None)
            if r:
                tn.append(t1); to.append(t0)
        a, b = statistics.median(tn), statistics.median(to)
        fl = 2 * M * n * k
        tot_new += a; tot_old += b
        print(f"{name} N={n:5d} K={k:5d}: {'persistent' if vs_pertile else 'K-blocked W'} {a * 1e3:7.1f} us {fl / a / 1e9:7.1f} TF/s (min {min(tn) * 1e3:6.1f}) | {'per-tile' if vs_pertile else 'row-major W'} {b * 1e3:7.1f} us {fl / b / 1e9:7.1f} TF/s"
              f" | x{b / a:.3f}  plan={K.gemm_nt_plan(M, n, k, n)}")
    print(f"sum: first {tot_new:.3f} ms, second {tot_old:.3f} ms (x{tot_old / tot_new:.3f}); per 12-layer step x12 = {12 * tot_new:.2f} vs {12 * tot_old:.2f} ms")


if __name__ == "__main__":
    main()
