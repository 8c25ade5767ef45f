"""A few launches of the variable-length attention kernels at the config-5 shape (ViT-H/14: N = 577, 16 heads of 80, batch 64) for
rocprofv3 --pmc passes (tools/pmc.sh tools/vl_prof.py <name> "<counters>" ...)."""
import torch

from vit_pytorch_amd import kernels as K
from vit_pytorch_amd.segments import Segments

dev = "cuda"
BF = torch.bfloat16
B, N, H, d = 64, 577, 16, 80
I = H * d
T = B * N
qkv = torch.randn(T, 3 * I, device=dev).to(BF); do = torch.randn(T, I, device=dev).to(BF)
o = torch.empty(T, I, dtype=BF, device=dev); lse = torch.empty(H, T, device=dev); delta = torch.empty(H, T, device=dev)
dqkv = torch.zeros_like(qkv)
sg = Segments([N] * B, [N] * B, torch.device(dev))
sn = 3 * I
q_, k_, v_ = K.hnd(qkv, d, sn), K.hnd(qkv, d, sn, offset=I), K.hnd(qkv, d, sn, offset=2 * I)
for _ in range(6):
    K.attn_varlen_fwd_bf16(q_, k_, v_, K.hnd(o, d, I), lse, sg.cu_q, sg.cu_k, sg.qblk_seg, sg.qblk_r0, sg.nqblk, T, H, d, d ** -0.5)
    K.attn_varlen_bwd_bf16(q_, k_, v_, K.hnd(o, d, I), K.hnd(do, d, I), lse, delta, K.hnd(dqkv, d, sn), K.hnd(dqkv, d, sn, offset=I),
                           K.hnd(dqkv, d, sn, offset=2 * I), sg.cu_q, sg.cu_k, sg.qblk_seg, sg.qblk_r0, sg.nqblk, sg.kblk_seg, sg.kblk_r0,
                           sg.nkblk, T, H, d, d ** -0.5)
torch.cuda.synchronize()
