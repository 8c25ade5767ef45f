"""Driver for rocprofv3 passes over the attention kernels at config-2 size (tools/pmc.sh tools/kprof_attn2.py <name> "<counters>" ...)."""
import torch
from vit_pytorch_amd import kernels as K
dev = "cuda"; BF = torch.bfloat16
B, N, H, d = 256, 197, 12, 64; I = H * d
qkv = torch.randn(B, N, 3 * I, device=dev).to(BF); o = torch.empty(B, N, I, dtype=BF, device=dev)
lse = torch.empty(B, H, N, device=dev); delta = torch.empty(B, H, N, device=dev)
do = torch.randn(B, N, I, device=dev).to(BF); dqkv = torch.empty_like(qkv)
sb, sh, sn = N * 3 * I, d, 3 * I
q_ = K.bhnd(qkv, sb, sh, sn); k_ = K.bhnd(qkv, sb, sh, sn, offset=I); v_ = K.bhnd(qkv, sb, sh, sn, offset=2 * I); o_ = K.bhnd(o, N * I, d, I)
for _ in range(4):
    K.attn_fwd_bf16(q_, k_, v_, o_, lse, B, H, N, d, d ** -0.5)
    K.attn_bwd_bf16(q_, k_, v_, o_, K.bhnd(do, N * I, d, I), lse, delta, K.bhnd(dqkv, sb, sh, sn), K.bhnd(dqkv, sb, sh, sn, offset=I),
                    K.bhnd(dqkv, sb, sh, sn, offset=2 * I), B, H, N, d, d ** -0.5)
torch.cuda.synchronize()
