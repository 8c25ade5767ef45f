"""Why does a GEMM take longer inside the training step than alone?  (round 5: FF1 350-370 us in the step, 290-300 us alone on the same
kind of box; QKV 185 vs 160, FF2 259 vs 225.)  The FF1 launch timed with HIP events inside loops that add its neighbours one by one:
python tools/instep_ab.py"""
import torch
from vit_pytorch_amd import ops, kernels as K, _lib as L

dev = "cuda"; BF = torch.bfloat16
M, D, F = 50432, 768, 3072
torch.manual_seed(0)
x = torch.randn(M, D, device=dev).to(BF)
lw = torch.ones(D, device=dev, dtype=BF); lb = torch.zeros(D, device=dev, dtype=BF)
w1 = ((torch.rand(F, D, device=dev) - 0.5) * 0.072).to(BF); b1 = ((torch.rand(F, device=dev) - 0.5) * 0.072).to(BF)
w2 = ((torch.rand(D, F, device=dev) - 0.5) * 0.036).to(BF); b2 = ((torch.rand(D, device=dev) - 0.5) * 0.036).to(BF)
a2 = torch.empty(M, D, device=dev, dtype=BF)
big = [torch.empty(64 * 1024 * 1024, device=dev, dtype=torch.float32) for _ in range(2)]        # 2 x 256 MiB: what a flush of the Infinity Cache touches


def ln():
    ops.ln_fwd(x, lw, lb, M, D, a2)


def loop(label, pre, post, iters=24, keep=False, flush=False):
    evs, kept = [], []
    for it in range(iters + 4):
        if flush:
            big[it & 1].add_(1.0)           # 512 MiB read + written: everything older leaves the caches
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        act, dg = ops.linear_fwd(a2, w1, b1, M, gelu=True, save_dg=True)
        e1.record()
        out = post(act)
        if keep:
            kept.append((act, dg, out))
            if len(kept) > 12:
                kept.clear()
        if it >= 4:
            evs.append((e0, e1))
    torch.cuda.synchronize()
    t = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in evs)
    print(f"{label:78s} FF1 median {t[len(t) // 2]:7.1f} us  (min {t[0]:7.1f}, max {t[-1]:7.1f})", flush=True)


nop = lambda *a: None
ff2 = lambda act: ops.linear_fwd(act, w2, b2, M, resid=x)
ln()
loop("FF1 alone (operand a2 and the outputs' addresses warm)", nop, nop)
loop("LayerNorm -> FF1 (a2 freshly written)", ln, nop)
loop("FF1 -> FF2 + residual", nop, ff2)
loop("LayerNorm -> FF1 -> FF2", ln, ff2)
loop("LayerNorm -> FF1 -> FF2, outputs kept 12 iterations deep (fresh memory, like the step)", ln, ff2, keep=True)
loop("512 MiB of unrelated traffic -> FF1 alone", nop, nop, flush=True)
loop("512 MiB of unrelated traffic -> LayerNorm -> FF1 -> FF2, outputs kept", ln, ff2, keep=True, flush=True)
