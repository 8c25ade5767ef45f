"""GPU: fp8 (OCP e4m3) operand path of the NT GEMM (BASELINE config 5): the device-side amax / scale,
the quantiser against torch's float8_e4m3fn cast (bit exact), and the GEMM against an f64 product of the dequantised
operands (products of e4m3 values are exact in f32, so only the 16-bit output rounding remains)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from vit_pytorch_amd import _lib as L, kernels as K  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16
F8 = torch.float8_e4m3fn


def rel(a, b):
    a = a.detach().double().flatten(); b = b.detach().double().flatten()
    return ((a - b).norm() / b.norm()).item()


def test_amax_scale_and_quantiser_match_torch():
    g = torch.Generator(device=DEV).manual_seed(0)
    x = (torch.randn(1000, 768, device=DEV, generator=g) * 3).to(BF)
    sc = torch.empty(2, device=DEV)
    K.fp8_amax_scale(x, sc)
    amax = x.float().abs().max()
    assert torch.allclose(sc[0], 448.0 / amax, rtol=1e-6) and torch.allclose(sc[1], amax / 448.0, rtol=1e-6)
    out = torch.empty(x.shape, dtype=torch.uint8, device=DEV)
    K.quantize_fp8(x, out, scale_dev=sc)
    ref = (x.float() * sc[0]).clamp(-448, 448).to(F8)
    assert torch.equal(out.view(F8).float(), ref.float())          # same values (round to nearest even, saturating)
    K.quantize_fp8(x, out, scale=0.5)
    assert torch.equal(out.view(F8).float(), (x.float() * 0.5).clamp(-448, 448).to(F8).float())


@pytest.mark.parametrize("M,N,Kd", [(2048, 768, 768), (5000, 1280, 1280), (6304, 3072, 768)])
def test_gemm_nt_fp8_against_dequantised_product(M, N, Kd):
    g = torch.Generator(device=DEV).manual_seed(1)
    A = torch.randn(M, Kd, device=DEV, generator=g).to(BF)
    W = (torch.randn(N, Kd, device=DEV, generator=g) * Kd ** -0.5).to(BF)
    bias = torch.randn(N, device=DEV, generator=g).to(BF)
    sa = torch.empty(2, device=DEV); sw = torch.empty(2, device=DEV)
    K.fp8_amax_scale(A, sa); K.fp8_amax_scale(W, sw)
    A8 = torch.empty(M, Kd, dtype=torch.uint8, device=DEV); W8 = torch.empty(N, Kd, dtype=torch.uint8, device=DEV)
    K.quantize_fp8(A, A8, scale_dev=sa); K.quantize_fp8(W, W8, scale_dev=sw)
    alpha = float(sa[1] * sw[1])
    ref = (A8.view(F8).double() @ W8.view(F8).double().t()) * alpha
    C = torch.empty(M, N, dtype=BF, device=DEV)
    K.gemm_nt_fp8(A8, Kd, W8, Kd, C, N, M, N, Kd, alpha)
    assert rel(C, ref) < 4e-3                                       # bf16 output rounding only
    assert rel(C, A.double() @ W.double().t()) < 6e-2               # and the quantisation itself costs a few percent
    aux = torch.empty(M, N, dtype=BF, device=DEV)
    K.gemm_nt_fp8(A8, Kd, W8, Kd, C, N, M, N, Kd, alpha, L.EPI_BIAS_GELU, bias=bias, aux=aux)
    pre = ref + bias.double()
    assert rel(aux, pre) < 4e-3 and rel(C, torch.nn.functional.gelu(pre)) < 4e-3
    resid = torch.randn(M, N, device=DEV, generator=g)
    out = torch.empty(M, N, device=DEV)
    K.gemm_nt_fp8(A8, Kd, W8, Kd, out, N, M, N, Kd, alpha, L.EPI_RESID, bias=bias, resid=resid)
    assert rel(out, resid.double() + pre) < 1e-5
    with pytest.raises(L.VitkError):
        K.gemm_nt_fp8(A8[:256], Kd, W8, Kd, C, N, 256, N, Kd, alpha)    # small M: not served by this prototype
