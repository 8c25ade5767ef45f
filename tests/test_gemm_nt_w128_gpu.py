"""GPU: the four-wave NT GEMM (csrc/gemm_nt_w128.hip) -- nn.Linear forward and dX of vit.py:20,23,44,47.

The product path splits the rows of a call: the four-wave kernel takes the full 256-row tiles of whole rounds, the 8-wave persistent
kernel the remaining rows.  Checked here:
  * against float64 products of the same operands (the same gates as the 8-wave kernel's tests);
  * BIT-IDENTICAL to the 8-wave kernel alone (VITK_NT_W128=0, read per call) for every epilogue -- same products, same accumulation
    order per output element, same epilogue arithmetic -- K-blocked and row-major W, strided A, in-place residual;
  * the column-sum partial rows of both launches add up to colsum(C);
  * run-to-run bit identity (a race in the LDS ring, in the counted waits or in the cross-tile prefetch shows as a changing result);
  * the IEEE-half library flavour;
  * shapes it must refuse fall through to the 8-wave kernel unchanged."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from vit_pytorch_amd import kernels as K  # noqa: E402
from vit_pytorch_amd import _lib as L  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16


def rel(a, b):
    a = a.double().flatten(); b = b.double().flatten()
    n = b.norm().item()
    return (a - b).norm().item() / (n if n > 0 else 1.0)


def rnd(*shape, dtype=torch.float32, seed=0):
    g = torch.Generator(device="cpu"); g.manual_seed(seed)
    return torch.randn(*shape, generator=g).to(dtype).to(DEV)


class eight_wave:
    """with eight_wave(): the 8-wave persistent kernel alone (the dispatcher reads VITK_NT_W128 on every call)."""

    def __enter__(self):
        self.prev = os.environ.get("VITK_NT_W128")
        os.environ["VITK_NT_W128"] = "0"

    def __exit__(self, *exc):
        if self.prev is None:
            os.environ.pop("VITK_NT_W128", None)
        else:
            os.environ["VITK_NT_W128"] = self.prev


def run_all(M, N, Kd, A, lda, W, ldw, bias, r32, r16, h, T=BF):
    """every epilogue of the 16-bit NT GEMM; returns {name: tensor}"""
    out = {}
    C = torch.full((M, N), float("nan"), dtype=T, device=DEV)
    K.gemm_nt_bf16(A, lda, W, ldw, C, N, M, N, Kd)
    out["none"] = C.clone()
    C.fill_(float("nan"))
    K.gemm_nt_bf16(A, lda, W, ldw, C, N, M, N, Kd, L.EPI_BIAS, bias=bias)
    out["bias"] = C.clone()
    aux = torch.full((M, N), float("nan"), dtype=T, device=DEV)
    C.fill_(float("nan"))
    K.gemm_nt_bf16(A, lda, W, ldw, C, N, M, N, Kd, L.EPI_BIAS_GELU, bias=bias, aux=aux)
    out["gelu"] = C.clone(); out["pre"] = aux.clone()
    aux.fill_(float("nan")); C.fill_(float("nan"))
    K.gemm_nt_bf16(A, lda, W, ldw, C, N, M, N, Kd, L.EPI_BIAS_GELU_DG, bias=bias, aux=aux)
    out["gelu_dg"] = C.clone(); out["dg"] = aux.clone()
    aux8 = torch.full((M, N), 255, dtype=torch.uint8, device=DEV); C.fill_(float("nan"))       # the factor as 8-bit codes (codes stop at 253)
    K.gemm_nt_bf16(A, lda, W, ldw, C, N, M, N, Kd, L.EPI_BIAS_GELU_DG8, bias=bias, aux=aux8)
    out["gelu_dg8"] = C.clone(); out["dg8"] = aux8.clone()
    o32 = torch.full((M, N), float("nan"), device=DEV)
    K.gemm_nt_bf16(A, lda, W, ldw, o32, N, M, N, Kd, L.EPI_RESID, bias=bias, resid=r32)
    out["resid"] = o32
    C.fill_(float("nan"))
    K.gemm_nt_bf16(A, lda, W, ldw, C, N, M, N, Kd, L.EPI_RESID16, bias=bias, resid=r16)
    out["resid16"] = C.clone()
    R = K.gemm_nt_colsum_rows(M, N, Kd, N, T)
    part = torch.full((R * N,), float("nan"), device=DEV)
    C2 = torch.full((M, N), float("nan"), dtype=T, device=DEV)
    K.gemm_nt_bf16_gelu_bwd_colsum(A, lda, W, ldw, C2, N, M, N, Kd, h, part)
    out["gbwd"] = C2; out["gbwd_sum"] = part.view(R, N).double().sum(0)
    part2 = torch.full((R * N,), float("nan"), device=DEV)
    C3 = torch.full((M, N), float("nan"), dtype=T, device=DEV)
    K.gemm_nt_bf16_mul_aux_colsum(A, lda, W, ldw, C3, N, M, N, Kd, h, part2)
    out["mul"] = C3; out["mul_sum"] = part2.view(R, N).double().sum(0)
    part3 = torch.full((R * N,), float("nan"), device=DEV)
    C4 = torch.full((M, N), float("nan"), dtype=T, device=DEV)
    K.gemm_nt_bf16_mul_aux8_colsum(A, lda, W, ldw, C4, N, M, N, Kd, h.view(torch.uint8)[:, :N].contiguous(), part3)      # any byte is a code
    out["mul8"] = C4; out["mul8_sum"] = part3.view(R, N).double().sum(0)
    return out


# (M, N, K): rows left for the 8-wave kernel (partial last tile + the tail of a round) / none (M % 256 == 0, whole rounds) / one tile per
# workgroup and fewer / many tiles per workgroup / the ViT-B/16 layer widths at a quarter batch
SHAPES = [(6000, 768, 256), (4096, 256, 512), (65536, 256, 256), (12608, 768, 768), (12608, 3072, 768), (12608, 768, 3072), (2900, 2304, 320),
          (131072 + 1300, 512, 256)]


@pytest.mark.parametrize("M,N,Kd", SHAPES)
def test_four_wave_path_against_float64_and_bitwise_against_the_8_wave_kernel(M, N, Kd):
    A = rnd(M, Kd, dtype=BF, seed=1); W = (rnd(N, Kd, seed=2) * Kd ** -0.5).to(BF)
    bias = rnd(N, dtype=BF, seed=3); r32 = rnd(M, N, seed=4); r16 = rnd(M, N, dtype=BF, seed=6); h = rnd(M, N, dtype=BF, seed=5)
    Wp = torch.empty(K.pack_w_nt_bytes(N, Kd) // 2, dtype=BF, device=DEV)
    K.pack_w_nt(W, Kd, N, Kd, Wp, None)
    got = run_all(M, N, Kd, A, Kd, Wp, 0, bias, r32, r16, h)
    got_rm = run_all(M, N, Kd, A, Kd, W, Kd, bias, r32, r16, h)            # row-major W through the same kernel
    with eight_wave():
        old = run_all(M, N, Kd, A, Kd, Wp, 0, bias, r32, r16, h)
    for k in got:
        assert got[k].dtype == torch.uint8 or not torch.isnan(got[k]).any(), k
        if k.endswith("_sum"):      # the partial rows differ between the plans; their sums agree to f32 round-off
            assert rel(got[k], old[k]) < 1e-6, k
        else:
            assert torch.equal(got[k], old[k]), k
            assert torch.equal(got_rm[k], old[k]), k + " (row-major W)"
    ref = A.double() @ W.double().t()
    pre = ref + bias.double()
    assert rel(got["none"], ref) < 4e-3
    assert (got["none"].double() - ref.float().to(BF).double()).abs().max().item() <= 2 * 2 ** -8 * ref.abs().max().item()
    assert rel(got["bias"], pre) < 4e-3
    assert rel(got["pre"], pre) < 4e-3 and rel(got["gelu"], torch.nn.functional.gelu(pre)) < 4e-3
    assert rel(got["gelu_dg"], torch.nn.functional.gelu(pre)) < 4e-3
    prd = got["pre"].double().requires_grad_(True)
    torch.nn.functional.gelu(prd).sum().backward()
    assert rel(got["dg"], prd.grad) < 4e-3
    # the 8-bit factor: same gelu output bit for bit; codes = rne(200 gelu'(pre)) + 27 up to one step where the f32 polynomial and the float64
    # derivative straddle a rounding boundary; decoded error <= 0.0025 (+ the polynomial's 2.2e-5)
    assert torch.equal(got["gelu_dg8"], got["gelu_dg"])
    code = got["dg8"].double()
    assert code.min().item() >= 1 and code.max().item() <= 253
    assert ((code - 27.0) * 0.005 - prd.grad).abs().max().item() <= 0.0025 + 1e-4
    assert (code - (torch.round(prd.grad * 200.0) + 27.0)).abs().max().item() <= 1
    assert rel(got["resid"], r32.double() + pre) < 1e-5
    assert rel(got["resid16"], r16.double() + pre) < 4e-3
    hd = h.double().requires_grad_(True)
    torch.nn.functional.gelu(hd).backward(ref)
    assert rel(got["gbwd"], hd.grad) < 4e-3
    assert rel(got["mul"], ref * h.double()) < 4e-3
    h8 = h.view(torch.uint8)[:, :N].double()
    assert rel(got["mul8"], ref * ((h8 - 27.0) * 0.005)) < 4e-3
    assert rel(got["mul8_sum"], got["mul8"].double().sum(0)) < 1e-5
    assert rel(got["gbwd_sum"], got["gbwd"].double().sum(0)) < 1e-5
    assert rel(got["mul_sum"], got["mul"].double().sum(0)) < 1e-5
    # run-to-run bit identity
    for _ in range(2):
        again = run_all(M, N, Kd, A, Kd, Wp, 0, bias, r32, r16, h)
        for k in got:
            if not k.endswith("_sum"):
                assert torch.equal(got[k], again[k]), k


def test_four_wave_strided_operand_and_in_place_residual():
    """The engine reads q|k|v slices of the merged projection by leading dimension and adds into the residual stream in place."""
    M, N, Kd, ld = 9000, 768, 256, 1024
    Abig = rnd(M, ld, dtype=BF, seed=11); W = (rnd(N, Kd, seed=12) * Kd ** -0.5).to(BF)
    A = Abig[:, 256:512]
    ref = A.double() @ W.double().t()
    C = torch.empty(M, N, dtype=BF, device=DEV)
    K.gemm_nt_bf16(A, ld, W, Kd, C, N, M, N, Kd)
    assert rel(C, ref) < 4e-3
    with eight_wave():
        C0 = torch.empty(M, N, dtype=BF, device=DEV)
        K.gemm_nt_bf16(A, ld, W, Kd, C0, N, M, N, Kd)
    assert torch.equal(C, C0)
    x = rnd(M, N, dtype=BF, seed=13); x0 = x.clone()
    K.gemm_nt_bf16(A, ld, W, Kd, x, N, M, N, Kd, L.EPI_RESID16, resid=x)          # in place
    assert rel(x, x0.double() + ref) < 4e-3
    xf = rnd(M, N, seed=14); xf0 = xf.clone()
    K.gemm_nt_bf16(A, ld, W, Kd, xf, N, M, N, Kd, L.EPI_RESID, resid=xf)
    assert rel(xf, xf0.double() + ref) < 1e-5


def test_four_wave_ieee_half_flavour():
    M, N, Kd = 6000, 512, 256
    H = torch.float16
    A = rnd(M, Kd, dtype=H, seed=21); W = (rnd(N, Kd, seed=22) * Kd ** -0.5).to(H)
    bias = rnd(N, dtype=H, seed=23); r32 = rnd(M, N, seed=24); r16 = rnd(M, N, dtype=H, seed=26); h = rnd(M, N, dtype=H, seed=25)
    got = run_all(M, N, Kd, A, Kd, W, Kd, bias, r32, r16, h, T=H)
    with eight_wave():
        old = run_all(M, N, Kd, A, Kd, W, Kd, bias, r32, r16, h, T=H)
    for k in got:
        if k.endswith("_sum"):
            assert rel(got[k], old[k]) < 1e-6, k
        else:
            assert torch.equal(got[k], old[k]), k
    ref = A.double() @ W.double().t()
    assert rel(got["none"], ref) < 6e-4
    assert rel(got["resid"], r32.double() + ref + bias.double()) < 1e-5


def test_column_sum_rows_follow_the_row_split():
    """vitk_gemm_nt_colsum_rows reports the partial rows of both launches; with the four-wave kernel off, the 8-wave plan's."""
    M, N, Kd = 50432, 768, 3072
    p = K.gemm_nt_plan(M, N, Kd, N)
    with eight_wave():
        assert K.gemm_nt_colsum_rows(M, N, Kd, N) == 2 * (p["tiles_m256"] + p["tiles_m128"])
    R = K.gemm_nt_colsum_rows(M, N, Kd, N)
    assert R >= 2 * (M // 256) and R <= 2 * ((M + 127) // 128)


@pytest.mark.parametrize("M,N,Kd", [(5000, 520, 256), (5000, 768, 96), (5000, 768, 192), (1000, 768, 768)])
def test_shapes_outside_the_four_wave_kernel_are_unchanged(M, N, Kd):
    """N % 256 != 0, K % 64 != 0 or K < 256, M < 1024: the earlier kernels serve them; results equal with the switch on and off."""
    A = rnd(M, Kd, dtype=BF, seed=31); W = (rnd(N, Kd, seed=32) * Kd ** -0.5).to(BF)
    C = torch.empty(M, N, dtype=BF, device=DEV); C0 = torch.empty(M, N, dtype=BF, device=DEV)
    K.gemm_nt_bf16(A, Kd, W, Kd, C, N, M, N, Kd)
    with eight_wave():
        K.gemm_nt_bf16(A, Kd, W, Kd, C0, N, M, N, Kd)
    assert torch.equal(C, C0)
    assert rel(C, A.double() @ W.double().t()) < 4e-3
