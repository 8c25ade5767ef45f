import torch, sys
sys.path.insert(0, "tools")
from vit_pytorch_amd import kernels as K, _lib as L
from kbench import timeit
dev="cuda"; BF=torch.bfloat16
M,D,F=50432,768,3072
x=torch.randn(M,D,device=dev); w=torch.ones(D,dtype=BF,device=dev); b=torch.zeros(D,dtype=BF,device=dev)
y=torch.empty(M,D,dtype=BF,device=dev); mean=torch.empty(M,device=dev); rstd=torch.empty(M,device=dev)
y8=torch.empty(M,D,dtype=torch.uint8,device=dev); sc=torch.tensor([10.0,0.1],device=dev); am=torch.zeros(64,dtype=torch.int32,device=dev)
print("ln_fwd bf16        %.3f ms" % timeit(lambda: K.layernorm_fwd(x,w,b,y,mean,rstd,M,D)))
print("ln_fwd +amax       %.3f ms" % timeit(lambda: K.layernorm_fwd(x,w,b,y,mean,rstd,M,D,amax64=am)))
print("ln_fwd +fp8 +amax  %.3f ms" % timeit(lambda: K.layernorm_fwd(x,w,b,y,mean,rstd,M,D,y8=y8,scale8=sc,amax64=am)))
A=torch.randn(M,D,device=dev).to(BF); W1=(torch.randn(F,D,device=dev)*D**-0.5).to(BF); b1=torch.randn(F,device=dev).to(BF)
act=torch.empty(M,F,dtype=BF,device=dev); pre=torch.empty(M,F,dtype=BF,device=dev); act8=torch.empty(M,F,dtype=torch.uint8,device=dev)
A8=torch.empty(M,D,dtype=torch.uint8,device=dev); W8=torch.empty(F,D,dtype=torch.uint8,device=dev)
K.quantize_fp8(A,A8,scale=8.0); K.quantize_fp8(W1,W8,scale=100.0)
print("ff1 bf16 gelu             %.3f ms" % timeit(lambda: K.gemm_nt_bf16(A,D,W1,D,act,F,M,F,D,L.EPI_BIAS_GELU,bias=b1,aux=pre)))
print("ff1 bf16 gelu +amax       %.3f ms" % timeit(lambda: K.gemm_nt_fp8_ex(A,D,W1,D,act,F,M,F,D,L.EPI_BIAS_GELU,a_is_fp8=False,bias=b1,aux=pre,c8_amax64=am)))
print("ff1 fp8 gelu              %.3f ms" % timeit(lambda: K.gemm_nt_fp8_ex(A8,D,W8,D,act,F,M,F,D,L.EPI_BIAS_GELU,a_is_fp8=True,bias=b1,aux=pre,alpha=1e-3)))
print("ff1 fp8 gelu +fp8out+amax %.3f ms" % timeit(lambda: K.gemm_nt_fp8_ex(A8,D,W8,D,act,F,M,F,D,L.EPI_BIAS_GELU,a_is_fp8=True,bias=b1,aux=pre,alpha=1e-3,alpha_a=sc[1:],alpha_w=sc[1:],c8=act8,c8_scale=sc,c8_amax64=am)))
W2=(torch.randn(D,F,device=dev)*F**-0.5).to(BF); W28=torch.empty(D,F,dtype=torch.uint8,device=dev); K.quantize_fp8(W2,W28,scale=100.0)
b2=torch.randn(D,device=dev).to(BF); res=torch.randn(M,D,device=dev); out=torch.empty(M,D,device=dev)
print("ff2 bf16 resid            %.3f ms" % timeit(lambda: K.gemm_nt_bf16(act,F,W2,F,out,D,M,D,F,L.EPI_RESID,bias=b2,resid=res)))
print("ff2 fp8 resid             %.3f ms" % timeit(lambda: K.gemm_nt_fp8_ex(act8,F,W28,F,out,D,M,D,F,L.EPI_RESID,a_is_fp8=True,bias=b2,resid=res,alpha=1e-3)))
Wq=(torch.randn(3*D,D,device=dev)*D**-0.5).to(BF); Wq8=torch.empty(3*D,D,dtype=torch.uint8,device=dev); K.quantize_fp8(Wq,Wq8,scale=100.0)
qkv=torch.empty(M,3*D,dtype=BF,device=dev)
print("qkv bf16                  %.3f ms" % timeit(lambda: K.gemm_nt_bf16(A,D,Wq,D,qkv,3*D,M,3*D,D)))
print("qkv fp8                   %.3f ms" % timeit(lambda: K.gemm_nt_fp8_ex(A8,D,Wq8,D,qkv,3*D,M,3*D,D,L.EPI_NONE,a_is_fp8=True,alpha=1e-3)))
