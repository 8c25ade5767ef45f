"""Times of the variable-length attention kernels (csrc/attention_varlen.hip) at the shapes of BASELINE config 5 (ViT-H/14: N = 577,
16 heads of 80, fixed-length batch seen as segments) and config 4 (NaViT: packed images of 16 .. 1600 tokens, 16 heads of 64).
One line per run; run once per library / geometry:   VITK_LIB=... VITK_ATTN_VL=1|2 python tools/vl_bench.py [batch]
Also checks forward and backward against a float64 reference on a small slice (first two segments, first two heads)."""
import os
import sys

import numpy as np
import torch

from vit_pytorch_amd import kernels as K
from vit_pytorch_amd.segments import Segments

dev = "cuda"
BF = torch.bfloat16


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def rel(a, b):
    a = a.double().flatten(); b = b.double().flatten()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def run(name, lens, H, d, scale):
    I = H * d
    T = sum(lens)
    g = torch.Generator(device=dev).manual_seed(5)
    qkv = torch.randn(T, 3 * I, generator=g, device=dev).to(BF)
    do = torch.randn(T, I, generator=g, device=dev).to(BF)
    o = torch.empty(T, I, dtype=BF, device=dev); lse = torch.empty(H, T, device=dev); delta = torch.empty(H, T, device=dev)
    dqkv = torch.zeros_like(qkv)
    sg = Segments(lens, lens, torch.device(dev))
    sn = 3 * I
    q_, k_, v_ = K.hnd(qkv, d, sn), K.hnd(qkv, d, sn, offset=I), K.hnd(qkv, d, sn, offset=2 * I)

    def fwd():
        K.attn_varlen_fwd_bf16(q_, k_, v_, K.hnd(o, d, I), lse, sg.cu_q, sg.cu_k, sg.qblk_seg, sg.qblk_r0, sg.nqblk, T, H, d, scale)

    def bwd():
        K.attn_varlen_bwd_bf16(q_, k_, v_, K.hnd(o, d, I), K.hnd(do, d, I), lse, delta, K.hnd(dqkv, d, sn), K.hnd(dqkv, d, sn, offset=I),
                               K.hnd(dqkv, d, sn, offset=2 * I), sg.cu_q, sg.cu_k, sg.qblk_seg, sg.qblk_r0, sg.nqblk, sg.kblk_seg, sg.kblk_r0,
                               sg.nkblk, T, H, d, scale)

    tf = timeit(fwd)
    tb = timeit(bwd)
    # float64 check on the first two segments, first two heads
    errs = []
    start = 0
    for n in lens[:2]:
        for h in range(min(H, 2)):
            sl = slice(start, start + n)
            qd = qkv[sl, h * d:(h + 1) * d].double().requires_grad_(True)
            kd = qkv[sl, I + h * d:I + (h + 1) * d].double().requires_grad_(True)
            vd = qkv[sl, 2 * I + h * d:2 * I + (h + 1) * d].double().requires_grad_(True)
            ref = torch.softmax(qd @ kd.T * scale, -1) @ vd
            ref.backward(do[sl, h * d:(h + 1) * d].double())
            errs.append((rel(o[sl, h * d:(h + 1) * d], ref), rel(dqkv[sl, h * d:(h + 1) * d], qd.grad),
                         rel(dqkv[sl, I + h * d:I + (h + 1) * d], kd.grad), rel(dqkv[sl, 2 * I + h * d:2 * I + (h + 1) * d], vd.grad)))
        start += n
    e = np.max(np.asarray(errs), axis=0)
    fl = 4.0 * sum(n * n for n in lens) * d * H        # QK^T + PV of the forward
    print(f"[{os.path.basename(os.environ.get('VITK_LIB', 'libvitk.so'))} VL={os.environ.get('VITK_ATTN_VL', '-')}] {name}: fwd {tf:.0f} us ({fl / tf * 1e-6:.0f} TF/s)"
          f"  bwd {tb:.0f} us ({2.5 * fl / tb * 1e-6:.0f} TF/s)   err o {e[0]:.1e} dq {e[1]:.1e} dk {e[2]:.1e} dv {e[3]:.1e}", flush=True)


B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
run(f"ViT-H/14 B={B} N=577 H=16 d=80" + (f" DBG={os.environ['VITK_VL_DBG']}" if os.environ.get("VITK_VL_DBG") else ""), [577] * B, 16, 80, 80 ** -0.5)
if os.environ.get("VL_BENCH_ONLY") == "h14":
    sys.exit(0)
rng = np.random.RandomState(0)
lens = [int(a * b) for a, b in zip(rng.randint(4, 41, size=48), rng.randint(4, 41, size=48))]
run(f"NaViT {len(lens)} images {sum(lens)} tokens H=16 d=64", lens, 16, 64, 1.0)
run(f"ViT-B/16 B={B} N=197 H=12 d=64 (varlen path)", [197] * B, 12, 64, 0.125)
