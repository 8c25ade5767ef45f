"""Does the FF1 GEMM's duration depend on the operand VALUES?  (round 5: inside the training step the launch takes 350-370 us, in
tools/nt_probe 280-292 us on the same kind of box -- same shape, same plan, fresh or reused buffers, seconds of sustained load: all level.)
The same launch on operands of different statistics: python tools/ff1_data_ab.py"""
import torch
from vit_pytorch_amd import kernels as K, _lib as L

dev = "cuda"; BF = torch.bfloat16
M, N, Kd = 50432, 3072, 768
g = torch.Generator(device=dev); g.manual_seed(0)


def time_us(fn, iters=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def run(label, A, W, bias):
    Wp = torch.empty(K.pack_w_nt_bytes(N, Kd) // 2, dtype=BF, device=dev)
    K.pack_w_nt(W, Kd, N, Kd, Wp, None)
    C = torch.empty(M, N, dtype=BF, device=dev); X8 = torch.empty(M, N, dtype=torch.uint8, device=dev); X16 = torch.empty(M, N, dtype=BF, device=dev)
    t0 = time_us(lambda: K.gemm_nt_bf16(A, Kd, Wp, 0, C, N, M, N, Kd, L.EPI_NONE))
    t8 = time_us(lambda: K.gemm_nt_bf16(A, Kd, Wp, 0, C, N, M, N, Kd, L.EPI_BIAS_GELU_DG8, bias=bias, aux=X8))
    t16 = time_us(lambda: K.gemm_nt_bf16(A, Kd, Wp, 0, C, N, M, N, Kd, L.EPI_BIAS_GELU_DG, bias=bias, aux=X16))
    print(f"{label:58s} plain stores {t0:7.1f} us | FF1 (8-bit factor) {t8:7.1f} us | FF1 (16-bit factor) {t16:7.1f} us", flush=True)


u = lambda *s: (torch.rand(*s, device=dev, generator=g) - 0.5)
n = lambda *s: torch.randn(*s, device=dev, generator=g)
bias0 = (u(N) * 0.5).to(BF)
run("zeros", torch.zeros(M, Kd, dtype=BF, device=dev), torch.zeros(N, Kd, dtype=BF, device=dev), torch.zeros(N, dtype=BF, device=dev))
run("probe-like: A uniform(-1, 1), W uniform(-0.05, 0.05)", (u(M, Kd) * 2).to(BF), (u(N, Kd) * 0.1).to(BF), bias0)
run("A normal(0, 1), W uniform(-0.036, 0.036) (nn.Linear init)", n(M, Kd).to(BF), (u(N, Kd) * 0.072).to(BF), (u(N) * 0.072).to(BF))
x = n(M, Kd) * 3 + 0.5
ln = torch.nn.functional.layer_norm(x, (Kd,)).to(BF)
run("A = LayerNorm output, W nn.Linear init", ln, (u(N, Kd) * 0.072).to(BF), (u(N) * 0.072).to(BF))
run("A normal(0, 1), W normal(0, 0.02)", n(M, Kd).to(BF), (n(N, Kd) * 0.02).to(BF), torch.zeros(N, dtype=BF, device=dev))
run("A normal(0, 4), W normal(0, 0.5)  (large pre-activations)", (n(M, Kd) * 4).to(BF), (n(N, Kd) * 0.5).to(BF), bias0)
run("A one value 1.0, W one value 0.01", torch.ones(M, Kd, dtype=BF, device=dev), torch.full((N, Kd), 0.01, dtype=BF, device=dev), bias0)
