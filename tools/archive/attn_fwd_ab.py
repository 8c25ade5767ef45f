"""Attention forward at ViT-B/16 batch 256 (B = 256, H = 12, N = 197, d = 64) under the resident / staggered option:
    VITK_ATTN_FWD_PERSIST=<workgroups per CU, 0 = one per item> VITK_ATTN_FWD_DELAY=<cycles> python tools/attn_fwd_ab.py"""
import os, torch
from vit_pytorch_amd import kernels as K
dev = "cuda"; BF = torch.bfloat16
B, H, N, d = 256, 12, 197, 64
I = H * d
torch.manual_seed(0)
qkv = torch.randn(B, N, 3 * I, device=dev).to(BF)
o = torch.empty(B, N, I, dtype=BF, device=dev); lse = torch.empty(B * H * N, device=dev)
sb, sh, sn = N * 3 * I, d, 3 * I
q_, k_, v_ = K.bhnd(qkv, sb, sh, sn), K.bhnd(qkv, sb, sh, sn, offset=I), K.bhnd(qkv, sb, sh, sn, offset=2 * I)
o_ = K.bhnd(o, N * I, d, I)
fn = lambda: K.attn_fwd_bf16(q_, k_, v_, o_, lse, B, H, N, d, d ** -0.5)
for _ in range(5): fn()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20)
print(f"[PERSIST={os.environ.get('VITK_ATTN_FWD_PERSIST', '0')} DELAY={os.environ.get('VITK_ATTN_FWD_DELAY', '-')}] {sorted(ts)[2] * 1e3:.1f} us  checksum {o.double().sum().item():.6e} {lse.double().sum().item():.6e}")
