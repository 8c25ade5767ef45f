"""A/B of the attention kernels at BASELINE config-2 size (B=256, H=12, N=197): python tools/attn_ab.py [B [N]]  (run once per setting of
VITK_ATTN_PIPE / VITK_LIB; prints forward / backward time with HIP events, 30 launches each after 5 warm-ups)."""
import sys
import torch
from vit_pytorch_amd import kernels as K

dev = "cuda"
BF = torch.bfloat16


def timeit(fn, iters=30, warm=5):
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


B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H, N, d = 12, (int(sys.argv[2]) if len(sys.argv) > 2 else 197), 64
I = H * d
qkv = torch.randn(B, N, 3 * I, device=dev).to(BF); o = torch.empty(B, N, I, dtype=BF, device=dev)
lse = torch.empty(B, H, N, device=dev); delta = torch.empty(B, H, N, device=dev)
do = torch.randn(B, N, I, device=dev).to(BF); dqkv = torch.empty_like(qkv)
sb, sh, sn = N * 3 * I, d, 3 * I
q_ = K.bhnd(qkv, sb, sh, sn); k_ = K.bhnd(qkv, sb, sh, sn, offset=I); v_ = K.bhnd(qkv, sb, sh, sn, offset=2 * I)
o_ = K.bhnd(o, N * I, d, I)
tf = timeit(lambda: K.attn_fwd_bf16(q_, k_, v_, o_, lse, B, H, N, d, d ** -0.5))
tb = timeit(lambda: K.attn_bwd_bf16(q_, k_, v_, o_, K.bhnd(do, N * I, d, I), lse, delta, K.bhnd(dqkv, sb, sh, sn), K.bhnd(dqkv, sb, sh, sn, offset=I),
                                    K.bhnd(dqkv, sb, sh, sn, offset=2 * I), B, H, N, d, d ** -0.5))
import os
print(f"VITK_ATTN_PIPE={os.environ.get('VITK_ATTN_PIPE', '(default)')} B={B} N={N}: fwd {tf:.1f} us  bwd {tb:.1f} us")
if int(os.environ.get("VITK_ATTN_DBG", "0")) & 24:
    # the fused backward's cycle stamps (workgroup 0): per-role totals in the (otherwise unused) delta buffer
    torch.cuda.synchronize()
    w = delta.view(-1)[:96].view(torch.int64).cpu().tolist()
    G = max(w[4], 1)
    print(f"  steps {G}; per step, cycles:  KV wave 2: loop head {w[0]/G:.0f}  first half {w[1]/G:.0f}  second half {w[2]/G:.0f}  barrier {w[3]/G:.0f}")
    print(f"  dQ wave 0: dQ {w[8]/G:.0f}  delta {w[9]/G:.0f}  barrier {w[10]/G:.0f};   producer: issue {w[16]/G:.0f}  wait {w[17]/G:.0f}  barrier {w[18]/G:.0f}")
    if int(os.environ.get("VITK_ATTN_DBG", "0")) & 16:
        print("  barrier wait per step, waves 0..15 (the shortest wait is the wave the others wait for):", " ".join(f"{x / G:.0f}" for x in w[20:36]))
