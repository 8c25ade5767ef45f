import torch
from vit_pytorch_amd import kernels as K
dev="cuda"; BF=torch.bfloat16
B,N,H,d=256,197,12,64; I=H*d
qkv=torch.randn(B,N,3*I,device=dev).to(BF); o=torch.empty(B,N,I,dtype=BF,device=dev)
lse=torch.empty(B,H,N,device=dev); delta=torch.empty(B,H,N,device=dev)
do=torch.randn(B,N,I,device=dev).to(BF); dqkv=torch.empty_like(qkv)
sb,sh,sn=N*3*I,d,3*I
q_=K.bhnd(qkv,sb,sh,sn); k_=K.bhnd(qkv,sb,sh,sn,offset=I); v_=K.bhnd(qkv,sb,sh,sn,offset=2*I); o_=K.bhnd(o,N*I,d,I)
for _ in range(3):
    K.attn_fwd_bf16(q_,k_,v_,o_,lse,B,H,N,d,d**-0.5)
    K.attn_bwd_bf16(q_,k_,v_,o_,K.bhnd(do,N*I,d,I),lse,delta,K.bhnd(dqkv,sb,sh,sn),K.bhnd(dqkv,sb,sh,sn,offset=I),K.bhnd(dqkv,sb,sh,sn,offset=2*I),B,H,N,d,d**-0.5)
# layernorm bwd too
M,D=B*N,768
x=torch.randn(M,D,device=dev); w=torch.ones(D,dtype=BF,device=dev); y=torch.randn(M,D,device=dev).to(BF)
mean=torch.zeros(M,device=dev); rstd=torch.ones(M,device=dev); gin=torch.randn(M,D,device=dev)
nblk=K.layernorm_bwd_blocks(M,D); partials=torch.empty(3*nblk*D,device=dev); dxf=torch.empty(M,D,device=dev); dxt=torch.empty(M,D,dtype=BF,device=dev)
for _ in range(3):
    K.layernorm_bwd(y,x,w,mean,rstd,gin,dxf,dxt,partials,True,M,D)
torch.cuda.synchronize()
