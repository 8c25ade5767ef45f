"""GPU: the Linear-layer entry points of ops.py (what engine.py calls for vit.py:20,23,44,47,102 and their autograd) on seeded random
(M, N, K) against float64 torch, across the dispatch boundaries of the GEMM kernels: M on both sides of 1024 (persistent NT kernel,
split-M weight gradients) and off the 128 / 256-row tiles, N and K off the 256 / 128 / 64 / 32 / 8 multiples, every epilogue the
engine uses (bias, bias + GELU with the saved pre-activation or the gelu' factor, float32 and 16-bit residual, GELU' + column sums,
dropout-free).  Tolerances (floating point): 16-bit operands with f32 accumulation, one rounding of the output -- 6e-3 relative L2
(2^-8 per element plus the reduction's round-off); f32 mode 1e-4."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from vit_pytorch_amd import ops  # noqa: E402

DEV = "cuda"
BF, F32 = torch.bfloat16, torch.float32


def rel(a, b):
    a = a.detach().double().flatten().cpu(); b = b.detach().double().flatten().cpu()
    n = b.norm().item()
    return (a - b).norm().item() / (n if n > 0 else 1.0)


def gelu64(x):
    return 0.5 * x * (1 + torch.erf(x / 2 ** 0.5))


def gelu_grad64(x):
    return 0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * np.pi) ** 0.5


def draw(seed):
    r = np.random.RandomState(4000 + seed)
    M = int(r.choice([1, 7, 100, 197, 256, 1000, 1023, 1024, 1025, 1182, 1280, 1300, 2049, 2304, 3000, 4097, 5000]))
    pick = lambda: int(r.choice([8, 16, 24, 40, 64, 72, 96, 128, 136, 192, 200, 256, 264, 320, 384, 512, 520, 768, 1000, 1024, 1280, 1536, 2304, 3072]))
    N, Kd = pick(), pick()
    if r.rand() < 0.15:
        N += 4          # off the 8-multiples: the generic coverage kernels
    if r.rand() < 0.15:
        Kd += 4
    return M, N, Kd


@pytest.mark.parametrize("dtype", [BF, F32])
@pytest.mark.parametrize("seed", range(36))
def test_linear_entry_points_random_shapes(seed, dtype):
    M, N, Kd = draw(seed)
    g = torch.Generator().manual_seed(seed)
    tol = 6e-3 if dtype == BF else 1e-4
    x = torch.randn(M, Kd, generator=g).to(dtype).to(DEV)
    W = (torch.randn(N, Kd, generator=g) / Kd ** 0.5).to(dtype).to(DEV)
    b = (0.1 * torch.randn(N, generator=g)).to(dtype).to(DEV)
    dy = torch.randn(M, N, generator=g).to(dtype).to(DEV)
    x64, W64, b64, dy64 = x.double(), W.double(), b.double(), dy.double()
    pre64 = x64 @ W64.T + b64
    tag = f"M={M} N={N} K={Kd} {dtype}"
    # y = x W^T + b ; without bias
    assert rel(ops.linear_fwd(x, W, b, M), pre64) <= tol, tag
    assert rel(ops.linear_fwd(x, W, None, M), x64 @ W64.T) <= tol, tag
    # bias + GELU, pre-activation saved
    act, pre = ops.linear_fwd(x, W, b, M, gelu=True)
    assert rel(pre, pre64) <= tol and rel(act, gelu64(pre.double())) <= tol, tag
    # bias + GELU with the gelu' factor saved instead (where the engine takes that pair)
    if ops.gelu_dg_ok(dtype, M, N, Kd):
        act2, dg = ops.linear_fwd(x, W, b, M, gelu=True, save_dg=True)
        assert rel(act2, gelu64(pre.double())) <= tol, tag
        if dg.dtype == torch.uint8:         # the factor as 8-bit fixed-point codes (bfloat16 models): |error| <= 0.0025 over its whole range
            assert ((dg.double() - 27.0) * 0.005 - gelu_grad64(pre.double())).abs().max().item() <= 0.0025 + 1e-4, tag
        else:
            assert rel(dg, gelu_grad64(pre.double())) <= tol, tag
    # residual epilogues: float32 stream, and the 16-bit stream where the engine runs it
    res = torch.randn(M, N, generator=g).to(DEV)
    out = ops.linear_fwd(x, W, b, M, resid=res)
    assert out.dtype == F32 and rel(out, res.double() + pre64) <= (tol if dtype == BF else 1e-4), tag
    if dtype == BF and ops._fast_nt(x, N, Kd) and ops._persistent_nt(M, N, Kd):
        r16 = res.to(BF)
        out16 = ops.linear_fwd(x, W, b, M, resid=r16)
        assert out16.dtype == BF and rel(out16, r16.double() + pre64) <= tol, tag
    # dX = dY W, alone and with the GELU backward (+ the bias gradient as a by-product where the kernel gives it)
    assert rel(ops.linear_dx(dy, W, M), dy64 @ W64) <= tol, tag
    hpre = torch.randn(M, Kd, generator=g).to(dtype).to(DEV)
    db = torch.empty(Kd, dtype=dtype, device=DEV)
    dxg, done = ops.linear_dx(dy, W, M, gelu_pre=hpre, db=db)
    ref = (dy64 @ W64) * gelu_grad64(hpre.double())
    assert rel(dxg, ref) <= 1.5 * tol, tag
    if done:
        assert rel(db, ref.sum(0)) <= 1.5 * tol + 1e-3, tag
    if dtype == BF and ops.gelu_dg_ok(dtype, M, Kd, N):
        fac = torch.rand(M, Kd, generator=g).to(dtype).to(DEV)
        dxm, done = ops.linear_dx(dy, W, M, gelu_dg=fac, db=db)
        refm = (dy64 @ W64) * fac.double()
        assert done and rel(dxm, refm) <= 1.5 * tol and rel(db, refm.sum(0)) <= 1.5 * tol + 1e-3, tag
        codes = torch.randint(0, 256, (M, Kd), generator=g, dtype=torch.uint8).to(DEV)         # the same on 8-bit codes of the factor
        dxm, done = ops.linear_dx(dy, W, M, gelu_dg=codes, db=db)
        refm = (dy64 @ W64) * ((codes.double() - 27.0) * 0.005)
        assert done and rel(dxm, refm) <= 1.5 * tol and rel(db, refm.sum(0)) <= 1.5 * tol + 1e-3, tag
    # dW = dY^T X, db = colsum(dY)
    dW = torch.empty(N, Kd, dtype=dtype, device=DEV); dbias = torch.empty(N, dtype=dtype, device=DEV)
    ops.linear_dw(dy, x, M, dW, dbias)
    assert rel(dW, dy64.T @ x64) <= tol and rel(dbias, dy64.sum(0)) <= tol, tag
