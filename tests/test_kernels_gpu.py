"""GPU: every libvitk kernel, called through the C-ABI, against a plain PyTorch reference of the
same op evaluated in fp32/fp64 on the same inputs.  Tolerances are stated per test: f32 kernels are
held to f32 round-off, bf16 kernels to bf16 output rounding (2^-8 relative) around an f32-accumulated
result."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from vit_pytorch_amd import kernels as K  # noqa: E402
from vit_pytorch_amd import _lib as L  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16
F32 = torch.float32


def rel(a, b):
    a = a.double().flatten(); b = b.double().flatten()
    n = b.norm().item()
    return (a - b).norm().item() / (n if n > 0 else 1.0)


def maxabs(a, b):
    return (a.double() - b.double()).abs().max().item()


def rnd(*shape, dtype=F32, scale=1.0, seed=None):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed if seed is not None else (hash(shape) & 0xffff))
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(DEV)


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("xdt,ydt,wdt", [(F32, F32, F32), (F32, BF, BF), (F32, F32, BF), (BF, BF, BF), (BF, F32, BF)])
@pytest.mark.parametrize("rows,D", [(37, 64), (1000, 768), (50, 1024), (9, 1280), (5, 3072), (6, 48), (98, 147), (25, 1323), (11, 3)])
def test_layernorm_fwd(xdt, ydt, wdt, rows, D):
    x = rnd(rows, D, dtype=xdt, seed=1) * 2 + 0.5
    w = (1 + 0.1 * rnd(D, seed=2)).to(wdt); b = (0.1 * rnd(D, seed=3)).to(wdt)
    y = torch.empty(rows, D, dtype=ydt, device=DEV)
    mean = torch.empty(rows, device=DEV); rstd = torch.empty(rows, device=DEV)
    K.layernorm_fwd(x, w, b, y, mean, rstd, rows, D)
    ref = torch.nn.functional.layer_norm(x.double(), (D,), w.double(), b.double(), 1e-5)
    tol = 2e-6 if ydt == F32 else 4e-3
    assert rel(y, ref) < tol
    assert rel(mean, x.double().mean(-1)) < 1e-5
    assert rel(rstd, 1 / torch.sqrt(x.double().var(-1, unbiased=False) + 1e-5)) < 1e-5


@pytest.mark.parametrize("rows,D", [(50432, 768), (1577, 768), (1024, 1024), (25216 + 1, 1024), (4616 + 3, 1280), (2049, 1280)])
@pytest.mark.parametrize("bias", [True, False])
def test_layernorm_fwd_16bit_rows_two_per_wave(rows, D, bias):
    """ln_fwd16_kernel (round 4: 16-bit x, y, w of 768 / 1024 / 1280 columns, identity maps: two rows per wave, 16-byte accesses) against
    float64, row counts that end inside a pair of rows / a block, NaN behind the last row (a half past the end must store nothing)."""
    x = rnd(rows, D, dtype=BF, seed=21) * 2 + 0.5
    w = (1 + 0.1 * rnd(D, seed=22)).to(BF); b = (0.1 * rnd(D, seed=23)).to(BF) if bias else None
    ybuf = torch.full((rows + 2, D), float("nan"), dtype=BF, device=DEV); y = ybuf[:rows]
    mbuf = torch.full((rows + 2,), float("nan"), device=DEV); rbuf = torch.full((rows + 2,), float("nan"), device=DEV)
    K.layernorm_fwd(x, w, b, y, mbuf[:rows], rbuf[:rows], rows, D)
    ref = torch.nn.functional.layer_norm(x.double(), (D,), w.double(), b.double() if bias else None, 1e-5)
    assert rel(y, ref) < 4e-3
    assert rel(mbuf[:rows], x.double().mean(-1)) < 1e-5
    assert rel(rbuf[:rows], 1 / torch.sqrt(x.double().var(-1, unbiased=False) + 1e-5)) < 1e-5
    assert torch.isnan(ybuf[rows:]).all() and torch.isnan(mbuf[rows:]).all() and torch.isnan(rbuf[rows:]).all()
    # the same values as the general kernel's, element for element (same arithmetic per element; the sums are formed in another order)
    y2 = torch.empty(rows, D, dtype=BF, device=DEV); m2 = torch.empty(rows, device=DEV); r2 = torch.empty(rows, device=DEV)
    K.layernorm_fwd(x.float(), w, b, y2, m2, r2, rows, D)          # f32 input: the general kernel
    assert (y.float() - y2.float()).abs().max().item() <= 2 * 2 ** -8 * ref.abs().max().item()


def test_layernorm_fwd_rowmaps_and_posadd():
    B, Np, D = 3, 6, 64
    N = Np + 1
    x = rnd(B * Np, D, dtype=BF, seed=4)
    w = rnd(D, dtype=BF, seed=5); b = rnd(D, dtype=BF, seed=6)
    pos = rnd(N, D, dtype=BF, seed=7)
    out = torch.zeros(B, N, D, device=DEV)
    mean = torch.empty(B * Np, device=DEV); rstd = torch.empty(B * Np, device=DEV)
    K.layernorm_fwd(x, w, b, out, mean, rstd, B * Np, D, omap=L.RowMap(Np, N, 1), add=pos, add_group=Np, add_off=1)
    ref = torch.nn.functional.layer_norm(x.float(), (D,), w.float(), b.float()).view(B, Np, D) + pos.float()[1:]
    assert rel(out[:, 1:], ref) < 2e-6
    assert out[:, 0].abs().max().item() == 0
    # read only the cls rows of a (B,N,D) tensor
    xs = rnd(B, N, D, seed=8)
    y = torch.empty(B, D, dtype=BF, device=DEV)
    m2 = torch.empty(B, device=DEV); r2 = torch.empty(B, device=DEV)
    K.layernorm_fwd(xs, w, b, y, m2, r2, B, D, imap=L.RowMap(1, N, 0))
    ref = torch.nn.functional.layer_norm(xs[:, 0], (D,), w.float(), b.float())
    assert rel(y, ref) < 4e-3


@pytest.mark.parametrize("dydt,xdt,wdt", [(F32, F32, F32), (BF, F32, BF), (BF, BF, BF), (F32, F32, BF)])
@pytest.mark.parametrize("rows,D", [(333, 768), (40, 64), (7, 1280), (2100, 256), (3100, 1024), (2600, 1280), (6272, 147), (700, 1323), (33, 5)])
def test_layernorm_bwd(dydt, xdt, wdt, rows, D):
    x = rnd(rows, D, dtype=xdt, seed=11) * 1.5 + 0.3
    dy = rnd(rows, D, dtype=dydt, seed=12)
    w = (1 + 0.2 * rnd(D, seed=13)).to(wdt); b = (0.1 * rnd(D, seed=14)).to(wdt)
    gin = rnd(rows, D, seed=15)
    xd = x.double().requires_grad_(True); wd = w.double().requires_grad_(True); bd = b.double().requires_grad_(True)
    yref = torch.nn.functional.layer_norm(xd, (D,), wd, bd, 1e-5)
    yref.backward(dy.double())
    mean = x.double().mean(-1).float(); rstd = (1 / torch.sqrt(x.double().var(-1, unbiased=False) + 1e-5)).float()
    nblk = K.layernorm_bwd_blocks(rows, D)
    partials = torch.empty(3 * nblk * D, device=DEV)
    dxf = torch.empty(rows, D, device=DEV)
    dxt = torch.empty(rows, D, dtype=wdt, device=DEV)
    K.layernorm_bwd(dy, x, w, mean, rstd, gin, dxf, dxt, partials, True, rows, D)
    dw = torch.empty(D, dtype=wdt, device=DEV); db = torch.empty(D, dtype=wdt, device=DEV); dc = torch.empty(D, device=DEV)
    K.colsum_partials(partials, nblk, D, D, dw)
    K.colsum_partials(partials[nblk * D:], nblk, D, D, db)
    K.colsum_partials(partials[2 * nblk * D:], nblk, D, D, dc)
    dx_ref = xd.grad + gin.double()
    assert rel(dxf, dx_ref) < 3e-6
    assert rel(dxt, dx_ref) < (3e-6 if wdt == F32 else 4e-3)
    tolw = 1e-5 if wdt == F32 else 5e-3
    assert rel(dw, wd.grad) < tolw and rel(db, bd.grad) < tolw
    assert rel(dc, dx_ref.sum(0)) < 1e-4


@pytest.mark.parametrize("xdt", [F32, BF])
@pytest.mark.parametrize("rows,D", [(2100, 768), (40, 64), (3100, 1024), (2600, 1280), (900, 256)])
def test_layernorm_bwd_16bit_stream(xdt, rows, D):
    """vitk_layernorm_bwd_s16: the stream gradient comes in and leaves in the 16-bit type (dx_t = dx + gin, no float32 output) --
    the specialised kernel at D = 768 / 1024 / 1280 and the general one; against float64 and against the float32-stream kernel
    fed the same (widened) gin: the two may differ only by the final rounding."""
    x = rnd(rows, D, dtype=xdt, seed=21) * 1.5 + 0.3
    dy = rnd(rows, D, dtype=BF, seed=22)
    w = (1 + 0.2 * rnd(D, seed=23)).to(BF)
    gin = rnd(rows, D, dtype=BF, seed=25)
    xd = x.double().requires_grad_(True); wd = w.double().requires_grad_(True)
    torch.nn.functional.layer_norm(xd, (D,), wd, None, 1e-5).backward(dy.double())
    mean = x.double().mean(-1).float(); rstd = (1 / torch.sqrt(x.double().var(-1, unbiased=False) + 1e-5)).float()
    nblk = K.layernorm_bwd_blocks(rows, D)
    partials = torch.empty(3 * nblk * D, device=DEV)
    dxt = torch.empty(rows, D, dtype=BF, device=DEV)
    K.layernorm_bwd_s16(dy, x, w, mean, rstd, gin, dxt, partials, True, rows, D)
    dw = torch.empty(D, dtype=BF, device=DEV); db = torch.empty(D, dtype=BF, device=DEV); dc = torch.empty(D, device=DEV)
    K.layernorm_bwd_finalize(partials, nblk, D, dw, db, dc, K.dt(w))
    dx_ref = xd.grad + gin.double()
    assert rel(dxt, dx_ref) < 4e-3
    assert rel(dw, wd.grad) < 5e-3 and rel(db, dy.double().sum(0)) < 5e-3
    assert rel(dc, dx_ref.sum(0)) < 1e-4
    # the float32-stream kernel on the same inputs: same values before the last rounding
    p2 = torch.empty(3 * nblk * D, device=DEV); dxf = torch.empty(rows, D, device=DEV); dxt2 = torch.empty(rows, D, dtype=BF, device=DEV)
    K.layernorm_bwd(dy, x, w, mean, rstd, gin.float(), dxf, dxt2, p2, True, rows, D)
    assert torch.equal(dxt, dxt2)
    # without an incoming gradient (the last LayerNorm of the stack)
    K.layernorm_bwd_s16(dy, x, w, mean, rstd, None, dxt, partials, True, rows, D)
    assert rel(dxt, xd.grad) < 4e-3


def test_layernorm_bwd_maps():
    # dy rows are the cls rows result (B rows), x rows are strided in (B,N,D), dx scattered to row 0 of each image
    B, N, D = 4, 5, 64
    xs = rnd(B, N, D, seed=21)
    dy = rnd(B, D, dtype=BF, seed=22)
    w = rnd(D, dtype=BF, seed=23)
    mean = xs[:, 0].mean(-1).contiguous(); rstd = (1 / torch.sqrt(xs[:, 0].var(-1, unbiased=False) + 1e-5)).contiguous()
    nblk = K.layernorm_bwd_blocks(B, D)
    partials = torch.empty(2 * nblk * D, device=DEV)
    dx = torch.zeros(B, N, D, device=DEV)
    m = L.RowMap(1, N, 0)
    K.layernorm_bwd(dy, xs, w, mean, rstd, None, dx, None, partials, False, B, D, xmap=m, dxmap=m)
    xd = xs[:, 0].double().requires_grad_(True)
    torch.nn.functional.layer_norm(xd, (D,), w.double(), None, 1e-5).backward(dy.double())
    assert rel(dx[:, 0], xd.grad) < 3e-6
    assert dx[:, 1:].abs().max().item() == 0


@pytest.mark.parametrize("rows,cols,xdt,odt", [(1000, 768, BF, BF), (257, 3072, BF, F32), (256, 197 * 64, F32, F32), (3, 8, F32, BF), (300, 10, BF, F32), (17, 7, F32, F32)])
def test_colsum(rows, cols, xdt, odt):
    x = rnd(rows, cols, dtype=xdt, seed=31)
    ws = torch.empty(K.colsum_ws_floats(rows, cols), device=DEV)
    out = torch.empty(cols, dtype=odt, device=DEV)
    K.colsum(x, rows, cols, cols, out, ws)
    assert rel(out, x.double().sum(0)) < (1e-5 if odt == F32 else 4e-3)
    K.colsum(x, rows, cols, cols, out, ws, accumulate=True)
    assert rel(out, 2 * x.double().sum(0)) < (1e-5 if odt == F32 else 8e-3)


# ---------------------------------------------------------------------------------------------
GEMM_SHAPES = [(1024, 256, 64), (2500, 768, 768), (1300, 520, 128), (128, 128, 32), (256, 384, 64), (197 * 3, 768, 768), (1000, 2304, 768), (130, 132, 96), (64, 64, 64), (50432 // 8, 768, 3072)]


@pytest.mark.parametrize("M,N,Kd", GEMM_SHAPES)
def test_gemm_nt_plain_and_bias(M, N, Kd):
    A = rnd(M, Kd, dtype=BF, seed=41); W = rnd(N, Kd, dtype=BF, seed=42) * (Kd ** -0.5)
    bias = rnd(N, dtype=BF, seed=43)
    ref = A.double() @ W.double().t()
    C = torch.empty(M, N, dtype=BF, device=DEV)
    K.gemm_nt_bf16(A, Kd, W, Kd, C, N, M, N, Kd)
    assert rel(C, ref) < 4e-3, "plain"
    # exactness up to bf16 output rounding: compare with the bf16-rounded f32 result
    assert maxabs(C, ref.float().to(BF)) <= 2 * 2 ** -8 * ref.abs().max().item()
    K.gemm_nt_bf16(A, Kd, W, Kd, C, N, M, N, Kd, L.EPI_BIAS, bias=bias)
    assert rel(C, ref + bias.double()) < 4e-3, "bias"


@pytest.mark.parametrize("M,N,Kd", [(197 * 2, 768, 768), (300, 3072, 768), (256, 768, 3072), (1100, 768, 768), (1576, 1032, 192)])
def test_gemm_nt_epilogues(M, N, Kd):
    A = rnd(M, Kd, dtype=BF, seed=44); W = rnd(N, Kd, dtype=BF, seed=45) * (Kd ** -0.5)
    bias = rnd(N, dtype=BF, seed=46)
    ref = A.double() @ W.double().t()
    # bias + gelu, saving the pre-activation
    C = torch.empty(M, N, dtype=BF, device=DEV); aux = torch.empty(M, N, dtype=BF, device=DEV)
    K.gemm_nt_bf16(A, Kd, W, Kd, C, N, M, N, Kd, L.EPI_BIAS_GELU, bias=bias, aux=aux)
    pre = ref + bias.double()
    assert rel(aux, pre) < 4e-3
    assert rel(C, torch.nn.functional.gelu(pre)) < 4e-3
    # residual (f32 stream), with and without bias
    resid = rnd(M, N, seed=47)
    out = torch.empty(M, N, device=DEV)
    K.gemm_nt_bf16(A, Kd, W, Kd, out, N, M, N, Kd, L.EPI_RESID, bias=bias, resid=resid)
    assert rel(out, resid.double() + pre) < 1e-5
    K.gemm_nt_bf16(A, Kd, W, Kd, out, N, M, N, Kd, L.EPI_RESID, resid=resid)
    assert rel(out, resid.double() + ref) < 1e-5
    # in-place residual (out aliases resid) is what the engine does
    r2 = resid.clone()
    K.gemm_nt_bf16(A, Kd, W, Kd, r2, N, M, N, Kd, L.EPI_RESID, resid=r2)
    assert rel(r2, resid.double() + ref) < 1e-5
    # gelu backward epilogue
    h = rnd(M, N, dtype=BF, seed=48)
    K.gemm_nt_bf16(A, Kd, W, Kd, C, N, M, N, Kd, L.EPI_GELU_BWD, aux=h)
    hd = h.double().requires_grad_(True)
    torch.nn.functional.gelu(hd).backward(ref)
    assert rel(C, hd.grad) < 4e-3


@pytest.mark.parametrize("M,N,Kd", [(1100, 768, 768), (1576, 1032, 192), (5 * 224 + 5, 3072, 768), (6304, 3072, 768)])
def test_gemm_nt_gelu_bwd_with_bias_gradient(M, N, Kd):
    """The GELU-backward product with colsum(C) (the first FeedForward Linear's bias gradient) as an epilogue by-product."""
    A = rnd(M, Kd, dtype=BF, seed=44); W = rnd(N, Kd, dtype=BF, seed=45) * (Kd ** -0.5)
    h = rnd(M, N, dtype=BF, seed=48)
    C0 = torch.empty(M, N, dtype=BF, device=DEV); C1 = torch.empty(M, N, dtype=BF, device=DEV)
    K.gemm_nt_bf16(A, Kd, W, Kd, C0, N, M, N, Kd, L.EPI_GELU_BWD, aux=h)
    R = K.gemm_nt_colsum_rows(M, N, Kd, N)
    assert R > 0
    part = torch.full((R * N,), float("nan"), device=DEV)
    K.gemm_nt_bf16_gelu_bwd_colsum(A, Kd, W, Kd, C1, N, M, N, Kd, h, part)
    assert torch.equal(C0, C1)
    db = torch.empty(N, dtype=BF, device=DEV)
    K.colsum_partials(part, R, N, N, db)
    ref = C1.double().sum(0)
    assert rel(part.view(R, N).double().sum(0), ref) < 1e-5          # f32 partials of the bf16-rounded C
    assert rel(db, ref) < 4e-3
    assert K.gemm_nt_colsum_rows(128, 128, 64, 128) == 0               # small shapes: not offered
    with pytest.raises(L.VitkError):
        K.gemm_nt_bf16_gelu_bwd_colsum(A[:128], Kd, W[:128], Kd, C1, 128, 128, 128, Kd, h, part)


@pytest.mark.parametrize("M,N,Kd", [(1200, 768, 768), (4200, 1032, 320)])
def test_f32_linear_on_the_mfma_kernels_via_bf16x3_split(M, N, Kd, monkeypatch):
    """f32 validation mode: ops.linear_fwd / linear_dx / linear_dw on float32 tensors run the PRODUCTION MFMA kernels (persistent NT,
    split-M TN) over three-term bfloat16 splits of both operands; the result must be f32-accurate (the bound here is 3e-6 of the
    float64 product; a plain bf16 product would sit at ~3e-3)."""
    from vit_pytorch_amd import ops
    x = rnd(M, Kd, seed=71); W = torch.nn.Parameter(rnd(N, Kd, seed=72) * Kd ** -0.5); b = rnd(N, seed=73); r = rnd(M, N, seed=74)
    assert K.gemm_nt_plan(M, N, 6 * Kd, N)["persistent"]
    ref = x.double() @ W.double().t()
    y = ops.linear_fwd(x, W, b, M)
    assert y.dtype == F32 and rel(y, ref + b.double()) < 3e-6
    yr = ops.linear_fwd(x, W, b, M, resid=r)
    assert rel(yr, ref + b.double() + r.double()) < 3e-6
    act, pre = ops.linear_fwd(x, W, b, M, gelu=True)
    assert rel(pre, ref + b.double()) < 3e-6 and rel(act, torch.nn.functional.gelu(ref + b.double())) < 3e-6
    dy = rnd(M, N, seed=75)
    dx = ops.linear_dx(dy, W, M)
    assert rel(dx, dy.double() @ W.double()) < 3e-6
    dW = torch.empty(N, Kd, device=DEV)
    ops.linear_dw(dy, x, M, dW)
    assert rel(dW, dy.double().t() @ x.double()) < (3e-6 if M >= 4096 else 1e-5)
    # and the same numbers (to round-off) with the mode switched off, i.e. on the VALU coverage kernel
    monkeypatch.setenv("VITK_F32_MFMA", "0")
    assert rel(ops.linear_fwd(x, W, b, M), y) < 3e-6


def test_gemm_nt_rejects_bad_shapes():
    A = rnd(64, 40, dtype=BF); W = rnd(64, 40, dtype=BF); C = torch.empty(64, 64, dtype=BF, device=DEV)
    with pytest.raises(L.VitkError):
        K.gemm_nt_bf16(A, 40, W, 40, C, 64, 64, 64, 40)  # K % 32 != 0


@pytest.mark.parametrize("M,N,Kd", [(4096, 256, 256), (5000, 768, 768), (6304, 2304, 768), (4500, 264, 520), (64, 128, 128), (1000, 768, 768), (197 * 16, 2304, 768), (333, 136, 72), (197 * 8, 768, 3072), (5000, 64, 256)])
@pytest.mark.parametrize("odt", [BF, F32])
def test_gemm_tn(M, N, Kd, odt):
    """Large shapes: gemm_tn_w128.hip (four waves with 128 x 128 wave tiles); the rest: the 128 x 128 kernel.  (The two 8-wave kernels of
    rounds 1-3 that VITK_TN_W128=0 / VITK_TN_DMA=1 used to select left the tree in round 5.)"""
    dY = rnd(M, N, dtype=BF, seed=51) * (M ** -0.5); X = rnd(M, Kd, dtype=BF, seed=52)
    ref = dY.double().t() @ X.double()
    splits = K.gemm_tn_splits(M, N, Kd)
    ws = torch.empty(splits * N * Kd, device=DEV)
    dW = torch.empty(N, Kd, dtype=odt, device=DEV)
    K.gemm_tn_bf16(dY, N, X, Kd, dW, Kd, M, N, Kd, ws, splits)
    assert rel(dW, ref) < (1e-5 if odt == F32 else 4e-3)
    K.gemm_tn_bf16(dY, N, X, Kd, dW, Kd, M, N, Kd, ws, splits, accumulate=True)
    assert rel(dW, 2 * ref) < (1e-5 if odt == F32 else 8e-3)


def test_gemm_tn_strided_operands():
    # dY is a column slice of a wider matrix (the merged dqkv buffer), X likewise
    M, N, Kd = 700, 128, 64
    big = rnd(M, 3 * N, dtype=BF, seed=53); xb = rnd(M, 2 * Kd, dtype=BF, seed=54)
    ref = big[:, N:2 * N].double().t() @ xb[:, Kd:].double()
    splits = 3
    ws = torch.empty(splits * N * Kd, device=DEV)
    dW = torch.empty(N, Kd, device=DEV)
    dYv = big[:, N:2 * N]; Xv = xb[:, Kd:]
    check = L.load().vitk_gemm_tn_bf16(dYv.data_ptr(), 3 * N, Xv.data_ptr(), 2 * Kd, dW.data_ptr(), L.F32, Kd, 0, M, N, Kd,
                                       ws.data_ptr(), splits, torch.cuda.current_stream().cuda_stream)
    assert check == 0
    assert rel(dW, ref) < 1e-5


@pytest.mark.parametrize("M,splits", [(4999, 5), (4130, 3), (8192, 9)])
def test_gemm_tn_w128_strided_ragged(M, splits):
    """The large-shape kernel on column slices of wider buffers (the merged dqkv gradient), with N and K that are not multiples of
    the 256-column tile and splits whose last one ends inside a 32-row step: rows past a split arrive as zeros through the buffer
    descriptor's range, columns past N / K only reach outputs that are never stored.  The buffers around the slices hold NaN."""
    N, Kd = 520, 264
    big = torch.full((M, N + 2 * 16), float("nan"), dtype=BF, device=DEV); xb = torch.full((M, Kd + 24), float("nan"), dtype=BF, device=DEV)
    big[:, 16:16 + N] = rnd(M, N, dtype=BF, seed=55) * (M ** -0.5); xb[:, 8:8 + Kd] = rnd(M, Kd, dtype=BF, seed=56)
    dYv = big[:, 16:16 + N]; Xv = xb[:, 8:8 + Kd]
    ref = dYv.double().t() @ Xv.double()
    ws = torch.empty(splits * N * Kd, device=DEV)
    dW = torch.empty(N, Kd, device=DEV)
    rc = L.load().vitk_gemm_tn_bf16(dYv.data_ptr(), big.stride(0), Xv.data_ptr(), xb.stride(0), dW.data_ptr(), L.F32, Kd, 0, M, N, Kd,
                                    ws.data_ptr(), splits, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    assert torch.isfinite(dW).all()
    assert rel(dW, ref) < 1e-5


@pytest.mark.parametrize("adt,bdt,cdt", [(F32, F32, F32), (BF, BF, BF), (BF, BF, F32)])
def test_gemm_generic_variants(adt, bdt, cdt):
    M, N, Kd = 70, 45, 33
    A = rnd(M, Kd, dtype=adt, seed=61); Bm = rnd(Kd, N, dtype=bdt, seed=62); bias = rnd(N, dtype=cdt, seed=63)
    C = torch.empty(M, N, dtype=cdt, device=DEV)
    tol = 1e-5 if cdt == F32 and adt == F32 else (1e-5 if cdt == F32 else 4e-3)
    # NN
    K.gemm_generic(K.mat(A, Kd, 1), K.mat(Bm, N, 1), K.mat(C, N, 1), M, N, Kd, bias=bias)
    assert rel(C, A.double() @ Bm.double() + bias.double()) < tol
    # NT: B given as (N, K) row-major
    Bt = Bm.t().contiguous()
    K.gemm_generic(K.mat(A, Kd, 1), K.mat(Bt, 1, Kd), K.mat(C, N, 1), M, N, Kd)
    assert rel(C, A.double() @ Bm.double()) < tol
    # TN: A given as (K, M) row-major, with alpha/beta
    At = A.t().contiguous()
    C0 = rnd(M, N, dtype=cdt, seed=64)
    C.copy_(C0)
    K.gemm_generic(K.mat(At, 1, M), K.mat(Bm, N, 1), K.mat(C, N, 1), M, N, Kd, alpha=0.5, beta=2.0)
    assert rel(C, 0.5 * (A.double() @ Bm.double()) + 2 * C0.double()) < (tol * 3)


def test_gemm_generic_batched_heads():
    # scores[b,h] = q[b,:,h,:] @ k[b,:,h,:]^T read in place from a merged (B, N, 3*H*d) tensor
    B, H, N, d = 2, 3, 19, 16
    I = H * d
    qkv = rnd(B, N, 3 * I, seed=65)
    S = torch.empty(B, H, N, N, device=DEV)
    K.gemm_generic(K.mat(qkv, 3 * I, 1, N * 3 * I, d), K.mat(qkv, 1, 3 * I, N * 3 * I, d, offset=I),
                   K.mat(S, N, 1, H * N * N, N * N), N, N, d, nb1=B, nb2=H)
    q = qkv[..., :I].view(B, N, H, d).permute(0, 2, 1, 3); k = qkv[..., I:2 * I].view(B, N, H, d).permute(0, 2, 1, 3)
    assert rel(S, q.double() @ k.double().transpose(-1, -2)) < 1e-5


# ---------------------------------------------------------------------------------------------
def _attn_ref(qkv, B, N, H, d, scale, do=None):
    I = H * d
    q, k, v = (qkv[..., i * I:(i + 1) * I].reshape(B, N, H, d).permute(0, 2, 1, 3).double() for i in range(3))
    q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
    s = (q @ k.transpose(-1, -2)) * scale
    p = torch.softmax(s, -1)
    o = p @ v
    om = o.permute(0, 2, 1, 3).reshape(B, N, I)
    lse = torch.logsumexp(s, -1)
    if do is None:
        return om.detach(), lse.detach()
    om.backward(do.double())
    g = [t.grad.permute(0, 2, 1, 3).reshape(B, N, I) for t in (q, k, v)]
    return om.detach(), lse.detach(), torch.cat(g, -1)


@pytest.mark.parametrize("B,H,N", [(2, 3, 197), (1, 2, 64), (3, 1, 50), (1, 4, 17), (2, 2, 256), (1, 1, 1)])
def test_attention_fwd_bwd(B, H, N):
    d = 64
    I = H * d
    scale = d ** -0.5
    qkv = rnd(B, N, 3 * I, dtype=BF, seed=71) * 1.5
    do = rnd(B, N, I, dtype=BF, seed=72)
    o = torch.empty(B, N, I, dtype=BF, device=DEV)
    lse = torch.empty(B, H, N, device=DEV)
    sb, sh, sn = N * 3 * I, d, 3 * I
    q_ = K.bhnd(qkv, sb, sh, sn); k_ = K.bhnd(qkv, sb, sh, sn, offset=I); v_ = K.bhnd(qkv, sb, sh, sn, offset=2 * I)
    o_ = K.bhnd(o, N * I, d, I)
    K.attn_fwd_bf16(q_, k_, v_, o_, lse, B, H, N, d, scale)
    oref, lref, gref = _attn_ref(qkv, B, N, H, d, scale, do)
    assert rel(o, oref) < 6e-3, rel(o, oref)
    assert maxabs(lse, lref) < 4e-3      # row sums are taken over the bf16-rounded probabilities (the ones P.V uses)
    dqkv = torch.zeros(B, N, 3 * I, dtype=BF, device=DEV)
    delta = torch.empty(B, H, N, device=DEV)
    K.attn_bwd_bf16(q_, k_, v_, o_, K.bhnd(do, N * I, d, I), lse, delta, K.bhnd(dqkv, sb, sh, sn),
                    K.bhnd(dqkv, sb, sh, sn, offset=I), K.bhnd(dqkv, sb, sh, sn, offset=2 * I), B, H, N, d, scale)
    for name, sl in (("dq", slice(0, I)), ("dk", slice(I, 2 * I)), ("dv", slice(2 * I, 3 * I))):
        r = rel(dqkv[..., sl], gref[..., sl])
        assert r < 1.2e-2, (name, r)


def test_attention_online_softmax_rescale_branch():
    # force the running max to jump late: one key (in the last 32-key step) dominates one query row
    B, H, N, d = 1, 1, 197, 64
    scale = d ** -0.5
    qkv = rnd(B, N, 3 * d, dtype=BF, seed=73) * 0.5
    qkv[0, 5, :d] = 4.0           # query 5
    qkv[0, 190, d:2 * d] = 4.0    # key 190 -> raw score 1024*scale = 128 for (5,190)
    o = torch.empty(B, N, d, dtype=BF, device=DEV); lse = torch.empty(B, H, N, device=DEV)
    sb, sh, sn = N * 3 * d, d, 3 * d
    K.attn_fwd_bf16(K.bhnd(qkv, sb, sh, sn), K.bhnd(qkv, sb, sh, sn, offset=d), K.bhnd(qkv, sb, sh, sn, offset=2 * d),
                    K.bhnd(o, N * d, d, d), lse, B, H, N, d, scale)
    oref, lref = _attn_ref(qkv, B, N, H, d, scale)
    assert torch.isfinite(o.float()).all()
    assert rel(o, oref) < 6e-3
    assert maxabs(lse, lref) < 2e-2


@pytest.mark.parametrize("dtype", [F32, BF])
def test_softmax_fwd_bwd(dtype):
    rows, cols, scale = 77, 197, 0.125
    s = rnd(rows, cols, dtype=dtype, seed=81) * 4
    p = torch.empty_like(s)
    K.softmax_fwd(s, p, rows, cols, scale)
    sd = s.double().requires_grad_(True)
    pref = torch.softmax(sd * scale, -1)
    assert rel(p, pref) < (2e-6 if dtype == F32 else 4e-3)
    dp = rnd(rows, cols, dtype=dtype, seed=82)
    ds = torch.empty_like(s)
    K.softmax_bwd(p, dp, ds, rows, cols, scale)
    # reference backward evaluated at the kernel's own p (so bf16 rounding of p is not counted twice)
    pd = p.double()
    dsref = scale * pd * (dp.double() - (dp.double() * pd).sum(-1, keepdim=True))
    assert rel(ds, dsref) < (3e-6 if dtype == F32 else 6e-3)


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [F32, BF])
@pytest.mark.parametrize("B,C,H,W,p1,p2", [(2, 3, 32, 32, 4, 4), (1, 3, 224, 224, 16, 16), (2, 1, 24, 32, 4, 8), (1, 3, 64, 64, 32, 32)])
def test_patchify(dtype, B, C, H, W, p1, p2):
    img = rnd(B, C, H, W, dtype=dtype, seed=91)
    h, w = H // p1, W // p2
    out = torch.empty(B * h * w, p1 * p2 * C, dtype=dtype, device=DEV)
    K.patchify(img, out, B, C, H, W, p1, p2)
    ref = img.reshape(B, C, h, p1, w, p2).permute(0, 2, 4, 3, 5, 1).reshape(B * h * w, p1 * p2 * C)
    assert torch.equal(out, ref)


@pytest.mark.parametrize("dtype", [F32, BF])
def test_gelu(dtype):
    x = rnd(1000, 64, dtype=dtype, seed=92) * 3
    y = torch.empty_like(x)
    K.gelu_fwd(x, y)
    xd = x.double().requires_grad_(True)
    yr = torch.nn.functional.gelu(xd)
    assert rel(y, yr) < (2e-6 if dtype == F32 else 4e-3)
    dy = rnd(1000, 64, dtype=dtype, seed=93)
    yr.backward(dy.double())
    dx = torch.empty_like(x)
    K.gelu_bwd(dy, x, dx)
    assert rel(dx, xd.grad) < (3e-6 if dtype == F32 else 4e-3)


def test_elementwise_extents_that_are_not_multiples_of_four():
    """T2T-ViT's 147- and 1323-wide layers (t2t.py:45): GELU and the residual add at odd extents, with and without a bias."""
    for dtype in (F32, BF):
        x = rnd(49, 147, dtype=dtype, seed=201) * 3
        y = torch.empty_like(x)
        K.gelu_fwd(x, y)
        xd = x.double().requires_grad_(True)
        yr = torch.nn.functional.gelu(xd)
        assert rel(y, yr) < (2e-6 if dtype == F32 else 4e-3)
        dy = rnd(49, 147, dtype=dtype, seed=202)
        yr.backward(dy.double())
        dx = torch.empty_like(x)
        K.gelu_bwd(dy, x, dx)
        assert rel(dx, xd.grad) < (3e-6 if dtype == F32 else 4e-3)
    rows, cols = 51, 1323
    a = rnd(rows, cols, seed=203); b = rnd(rows, cols, dtype=BF, seed=204); bias = rnd(cols, dtype=BF, seed=205)
    out = torch.empty(rows, cols, device=DEV)
    K.add_rows(a, b, bias, out, rows, cols)
    assert rel(out, a.double() + b.double() + bias.double()) < 1e-6
    ab = a.to(BF); outb = torch.empty(rows, cols, dtype=BF, device=DEV)
    K.add_rows(ab, b, None, outb, rows, cols)
    assert rel(outb, ab.double() + b.double()) < 4e-3


def test_add_rows_cast_cls_meanpool_transpose():
    rows, cols = 50, 64
    a = rnd(rows, cols, seed=94); b = rnd(rows, cols, dtype=BF, seed=95); bias = rnd(cols, dtype=BF, seed=96)
    out = torch.empty(rows, cols, device=DEV)
    K.add_rows(a, b, bias, out, rows, cols)
    assert rel(out, a.double() + b.double() + bias.double()) < 1e-6
    outb = torch.empty(rows, cols, dtype=BF, device=DEV)
    K.add_rows(a, b, None, outb, rows, cols)
    assert rel(outb, a.double() + b.double()) < 4e-3
    # cast
    x = rnd(1003, seed=97); y = torch.empty(1003, dtype=BF, device=DEV)
    K.cast(x, y)
    assert torch.equal(y, x.to(BF))
    z = torch.empty(1003, device=DEV)
    K.cast(y, z)
    assert torch.equal(z, y.float())
    # cls rows
    B, N, D = 3, 5, 64
    xs = torch.zeros(B, N, D, device=DEV); cls = rnd(1, D, dtype=BF, seed=98); pos = rnd(N, D, dtype=BF, seed=99)
    K.write_cls_rows(xs, cls, pos, B, N, D, 1)
    assert rel(xs[:, 0], (cls.float() + pos.float()[0:1]).expand(B, D)) < 1e-7
    assert xs[:, 1:].abs().max().item() == 0
    # mean pool
    xt = rnd(B, N, D, dtype=BF, seed=100); mp = torch.empty(B, D, dtype=BF, device=DEV)
    K.mean_pool_fwd(xt, mp, B, N, D)
    assert rel(mp, xt.double().mean(1)) < 4e-3
    dmp = rnd(B, D, dtype=BF, seed=101); dxs = torch.empty(B, N, D, device=DEV)
    K.mean_pool_bwd(dmp, dxs, B, N, D)
    assert rel(dxs, (dmp.double() / N).unsqueeze(1).expand(B, N, D)) < 1e-6
    # transpose
    w = rnd(70, 45, dtype=BF, seed=102); wt = torch.empty(45, 70, dtype=BF, device=DEV)
    K.transpose(w, wt, 70, 45)
    assert torch.equal(wt, w.t().contiguous())


def test_dropout_statistics_and_backward():
    n, p = 1 << 20, 0.1
    x = torch.ones(n, dtype=BF, device=DEV)
    y = torch.empty_like(x); mask = torch.empty(n, dtype=torch.uint8, device=DEV)
    K.dropout_fwd(x, y, mask, p, 1234, 0)
    keep = mask.float().mean().item()
    assert abs(keep - (1 - p)) < 3e-3
    assert rel(y, mask.float() / (1 - p)) < 4e-3
    y2 = torch.empty_like(x); m2 = torch.empty_like(mask)
    K.dropout_fwd(x, y2, m2, p, 1234, 0)
    assert torch.equal(mask, m2)                      # reproducible from (seed, offset)
    K.dropout_fwd(x, y2, m2, p, 1234, n)
    assert not torch.equal(mask, m2)                  # a different offset is a different stream
    dy = rnd(n, dtype=BF, seed=103); dx = torch.empty_like(dy)
    K.dropout_bwd(dy, mask, dx, p)
    assert rel(dx, dy.float() * mask.float() / (1 - p)) < 4e-3


# ---- NaViT / ViT-H data movement -------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [F32, BF])
def test_patchify_cpp_order_and_row_offset(dtype):
    C, H, W, p = 3, 24, 40, 8
    img = rnd(C, H, W, dtype=dtype, seed=201)
    n = (H // p) * (W // p)
    out = torch.zeros(n + 5, C * p * p, dtype=dtype, device=DEV)
    K.patchify_cpp(img, out, C, H, W, p, 3, C * p * p)
    ref = img.reshape(C, H // p, p, W // p, p).permute(1, 3, 0, 2, 4).reshape(n, C * p * p)   # (h w) (c p1 p2)
    assert torch.equal(out[3:3 + n], ref)
    assert out[:3].abs().max().item() == 0 and out[3 + n:].abs().max().item() == 0


@pytest.mark.parametrize("dtype", [F32, BF])
def test_gather_add2_and_csr_rowsum(dtype):
    T, D, nh, nw = 50, 64, 5, 7
    x = rnd(T, D, dtype=dtype, seed=202); A = rnd(nh, D, dtype=dtype, seed=203); Bm = rnd(nw, D, dtype=dtype, seed=204)
    g = torch.Generator().manual_seed(5)
    ia = torch.randint(0, nh, (T,), generator=g).to(torch.int32).to(DEV); ib = torch.randint(0, nw, (T,), generator=g).to(torch.int32).to(DEV)
    out = torch.empty_like(x)
    K.gather_add2(x, A, ia, Bm, ib, out, T, D)
    ref = x.double() + A.double()[ia.long()] + Bm.double()[ib.long()]
    assert rel(out, ref) < (1e-6 if dtype == F32 else 4e-3)
    # backward of the gather: dA[i] = sum of g over the tokens with ia == i (deterministic CSR form)
    gr = rnd(T, D, dtype=dtype, seed=205)
    order = torch.argsort(ia.long(), stable=True).to(torch.int32)
    ptr = torch.cat([torch.zeros(1, dtype=torch.long), torch.bincount(ia.long().cpu(), minlength=nh).cumsum(0)]).to(torch.int32).to(DEV)
    dA = torch.empty(nh, D, dtype=dtype, device=DEV)
    K.csr_rowsum(gr, ptr, order, dA, nh, D)
    refA = torch.zeros(nh, D, dtype=torch.float64, device=DEV).index_add_(0, ia.long(), gr.double())
    assert rel(dA, refA) < (1e-6 if dtype == F32 else 6e-3)


def test_copy_cols_pad_and_strip():
    rows, cols, pad = 37, 588, 608
    x = rnd(rows, cols, dtype=BF, seed=206)
    y = torch.full((rows, pad), 7.0, dtype=BF, device=DEV)
    K.copy_cols(x, cols, y, pad, rows, cols, pad)
    assert torch.equal(y[:, :cols], x) and y[:, cols:].abs().max().item() == 0
    z = torch.empty(rows, cols, dtype=BF, device=DEV)
    K.copy_cols(y, pad, z, cols, rows, cols, cols)
    assert torch.equal(z, x)


# ---------------------------------------------------------------------------------------------
# IEEE-half flavour (libvitk_f16.so): the same kernels, operands float16
F16 = torch.float16


@pytest.mark.parametrize("M,N,Kd", [(2500, 768, 768), (6304, 3072, 768), (130, 132, 96)])
def test_f16_gemms(M, N, Kd):
    A = rnd(M, Kd, seed=141).to(F16); W = (rnd(N, Kd, seed=142) * (Kd ** -0.5)).to(F16)
    bias = rnd(N, seed=143).to(F16)
    ref = A.double() @ W.double().t()
    C = torch.empty(M, N, dtype=F16, device=DEV); aux = torch.empty(M, N, dtype=F16, device=DEV)
    K.gemm_nt_bf16(A, Kd, W, Kd, C, N, M, N, Kd, L.EPI_BIAS_GELU, bias=bias, aux=aux)
    pre = ref + bias.double()
    assert rel(aux, pre) < 6e-4 and rel(C, torch.nn.functional.gelu(pre)) < 8e-4
    resid = rnd(M, N, seed=147); out = torch.empty(M, N, device=DEV)
    K.gemm_nt_bf16(A, Kd, W, Kd, out, N, M, N, Kd, L.EPI_RESID, bias=bias, resid=resid)
    assert rel(out, resid.double() + pre) < 1e-5
    if N % 8 == 0 and Kd % 8 == 0:
        dY = rnd(M, N, seed=148).to(F16)
        splits = K.gemm_tn_splits(M, N, Kd)
        ws = torch.empty(splits * N * Kd, device=DEV); dW = torch.empty(N, Kd, dtype=F16, device=DEV)
        K.gemm_tn_bf16(dY, N, A, Kd, dW, Kd, M, N, Kd, ws, splits)
        assert rel(dW, dY.double().t() @ A.double()) < 6e-4
    with pytest.raises(L.VitkError):                       # one 16-bit type per call
        K.gemm_nt_bf16(A, Kd, W.to(BF), Kd, C, N, M, N, Kd)


@pytest.mark.parametrize("B,H,N", [(2, 3, 197), (1, 2, 64), (2, 2, 256)])
def test_f16_attention_fwd_bwd(B, H, N):
    d = 64
    I = H * d
    scale = d ** -0.5
    qkv = (rnd(B, N, 3 * I, seed=171) * 1.5).to(F16)
    do = rnd(B, N, I, seed=172).to(F16)
    o = torch.empty(B, N, I, dtype=F16, device=DEV)
    lse = torch.empty(B, H, N, device=DEV); delta = torch.empty(B, H, N, device=DEV)
    sb, sh, sn = N * 3 * I, d, 3 * I
    q_ = K.bhnd(qkv, sb, sh, sn); k_ = K.bhnd(qkv, sb, sh, sn, offset=I); v_ = K.bhnd(qkv, sb, sh, sn, offset=2 * I)
    o_ = K.bhnd(o, N * I, d, I)
    K.attn_fwd_bf16(q_, k_, v_, o_, lse, B, H, N, d, scale)
    oref, lref, gref = _attn_ref(qkv, B, N, H, d, scale, do)
    assert rel(o, oref) < 1e-3 and maxabs(lse, lref) < 1e-3
    dqkv = torch.empty_like(qkv)
    K.attn_bwd_bf16(q_, k_, v_, o_, K.bhnd(do, N * I, d, I), lse, delta, K.bhnd(dqkv, sb, sh, sn),
                    K.bhnd(dqkv, sb, sh, sn, offset=I), K.bhnd(dqkv, sb, sh, sn, offset=2 * I), B, H, N, d, scale)
    assert rel(dqkv, gref) < 2e-3


# ---------------------------------------------------------------------------------------------
# fused dropout: decisions are a counter hash of (seed, row, col); vitk_dropout_keep materialises them for the reference
def test_dropout_keep_hash_statistics():
    keep = torch.empty(4096, 1000, dtype=torch.uint8, device=DEV)
    for p in (0.1, 0.5):
        K.dropout_keep(keep, 4096, 1000, p, 1234)
        f = keep.float()
        assert abs(f.mean().item() - (1 - p)) < 2e-3
        assert (f.mean(0) - (1 - p)).abs().max().item() < 0.04 and (f.mean(1) - (1 - p)).abs().max().item() < 0.07   # no dead rows / columns
        k2 = torch.empty_like(keep)
        K.dropout_keep(k2, 4096, 1000, p, 1235)
        assert abs((keep == k2).float().mean().item() - (p * p + (1 - p) ** 2)) < 3e-3                                # seeds are independent
    K.dropout_keep(keep, 4096, 1000, 0.0, 7)
    assert keep.min().item() == 1


@pytest.mark.parametrize("B,H,N", [(2, 3, 197), (1, 2, 64), (2, 2, 256)])
@pytest.mark.parametrize("r", ["1", "2"])
def test_attention_with_probability_dropout(B, H, N, r, monkeypatch):
    """nn.Dropout on the attention matrix inside the flash kernels (vit.py:60), forward and backward, against a float64
    reference that uses the very same keep decisions."""
    for v in ("VITK_ATTN_R_FWD", "VITK_ATTN_R_DQ", "VITK_ATTN_R_DKV"):
        monkeypatch.setenv(v, r)
    d, p, seed = 64, 0.2, 4321
    I = H * d
    scale = d ** -0.5
    qkv = rnd(B, N, 3 * I, dtype=BF, seed=271) * 1.5
    do = rnd(B, N, I, dtype=BF, seed=272)
    keep = torch.empty(B * H * N, N, dtype=torch.uint8, device=DEV)
    K.dropout_keep(keep, B * H * N, N, p, seed)
    mask = keep.view(B, H, N, N).double() / (1 - p)
    q, k, v = (qkv[..., i * I:(i + 1) * I].reshape(B, N, H, d).permute(0, 2, 1, 3).double().requires_grad_(True) for i in range(3))
    pm = torch.softmax((q @ k.transpose(-1, -2)) * scale, -1) * mask
    oref = (pm @ v).permute(0, 2, 1, 3).reshape(B, N, I)
    oref.backward(do.double())
    gref = torch.cat([t.grad.permute(0, 2, 1, 3).reshape(B, N, I) for t in (q, k, v)], -1)
    o = torch.empty(B, N, I, dtype=BF, device=DEV)
    lse = torch.empty(B, H, N, device=DEV); delta = torch.empty(B, H, N, device=DEV)
    sb, sh, sn = N * 3 * I, d, 3 * I
    q_ = K.bhnd(qkv, sb, sh, sn); k_ = K.bhnd(qkv, sb, sh, sn, offset=I); v_ = K.bhnd(qkv, sb, sh, sn, offset=2 * I)
    o_ = K.bhnd(o, N * I, d, I)
    K.attn_fwd_bf16(q_, k_, v_, o_, lse, B, H, N, d, scale, p, seed)
    assert rel(o, oref) < 8e-3, rel(o, oref)
    dqkv = torch.empty_like(qkv)
    K.attn_bwd_bf16(q_, k_, v_, o_, K.bhnd(do, N * I, d, I), lse, delta, K.bhnd(dqkv, sb, sh, sn),
                    K.bhnd(dqkv, sb, sh, sn, offset=I), K.bhnd(dqkv, sb, sh, sn, offset=2 * I), B, H, N, d, scale, p, seed)
    assert rel(dqkv, gref) < 1.5e-2, rel(dqkv, gref)


def _keep(rows, cols, p, seed):
    k = torch.empty(rows, cols, dtype=torch.uint8, device=DEV)
    K.dropout_keep(k, rows, cols, p, seed)
    return k.double() / (1 - p)


def test_gemm_epilogues_with_fused_dropout():
    M, N, Kd, p = 1300, 768, 768, 0.25
    A = rnd(M, Kd, dtype=BF, seed=344); W = rnd(N, Kd, dtype=BF, seed=345) * (Kd ** -0.5)
    bias = rnd(N, dtype=BF, seed=346)
    pre = A.double() @ W.double().t() + bias.double()
    mk = _keep(M, N, p, 99)
    resid = rnd(M, N, seed=347); out = torch.empty(M, N, device=DEV)
    K.gemm_nt_bf16_drop(A, Kd, W, Kd, out, N, M, N, Kd, L.EPI_RESID, p, 99, bias=bias, resid=resid)
    assert rel(out, resid.double() + pre * mk) < 1e-5
    C = torch.empty(M, N, dtype=BF, device=DEV); aux = torch.empty(M, N, dtype=BF, device=DEV)
    K.gemm_nt_bf16_drop(A, Kd, W, Kd, C, N, M, N, Kd, L.EPI_BIAS_GELU, p, 99, bias=bias, aux=aux)
    assert rel(aux, pre) < 4e-3 and rel(C, torch.nn.functional.gelu(pre) * mk) < 5e-3
    h = rnd(M, N, dtype=BF, seed=348)
    R = K.gemm_nt_colsum_rows(M, N, Kd, N)
    part = torch.empty(R * N, device=DEV)
    K.gemm_nt_bf16_drop(A, Kd, W, Kd, C, N, M, N, Kd, L.EPI_GELU_BWD, p, 99, aux=h, partials=part)
    hd = h.double().requires_grad_(True)
    (torch.nn.functional.gelu(hd) * mk).backward(A.double() @ W.double().t())
    assert rel(C, hd.grad) < 5e-3
    assert rel(part.view(R, N).double().sum(0), C.double().sum(0)) < 1e-5
    with pytest.raises(L.VitkError):                      # small shapes run on the 128-row kernel: no fused dropout there
        K.gemm_nt_bf16_drop(A[:256], Kd, W, Kd, out, N, 256, N, Kd, L.EPI_RESID, p, 99, bias=bias, resid=resid)


def test_layernorm_bwd_with_output_dropout():
    rows, D, p = 700, 768, 0.3
    x = rnd(rows, D, seed=411); dy = rnd(rows, D, dtype=BF, seed=412); w = (1 + 0.2 * rnd(D, seed=413)).to(BF); gin = rnd(rows, D, seed=415)
    mean = x.double().mean(-1).float(); rstd = (1 / torch.sqrt(x.double().var(-1, unbiased=False) + 1e-5)).float()
    nblk = K.layernorm_bwd_blocks(rows, D)
    partials = torch.empty(3 * nblk * D, device=DEV)
    dxf = torch.empty(rows, D, device=DEV); dxt = torch.empty(rows, D, dtype=BF, device=DEV)
    K.layernorm_bwd(dy, x, w, mean, rstd, gin, dxf, dxt, partials, True, rows, D, drop_p=p, drop_seed=77)
    xd = x.double().requires_grad_(True)
    torch.nn.functional.layer_norm(xd, (D,), w.double(), None, 1e-5).backward(dy.double())
    dx_ref = xd.grad + gin.double()
    mk = _keep(rows, D, p, 77)
    assert rel(dxf, dx_ref) < 3e-6                                         # the stream gradient is not masked
    assert rel(dxt, dx_ref * mk) < 4e-3                                    # the 16-bit copy (GEMM operand) is
    dc = torch.empty(D, device=DEV)
    K.colsum_partials(partials[2 * nblk * D:], nblk, D, D, dc)
    assert rel(dc, (dx_ref * mk).sum(0)) < 1e-4                            # and so is the bias gradient it yields


@pytest.mark.parametrize("B,H,W", [(256, 224, 224), (3, 48, 80), (5, 16, 16), (2, 224, 224)])
@pytest.mark.parametrize("bias", [True, False])
def test_patch_ln_gather_fwd_and_param_grads(B, H, W, bias):
    """vitk_patch_ln_fwd / vitk_patch_ln_bwd_params (round 4: Rearrange + LayerNorm(patch_dim) with the gather in the load, vit.py:100-101)
    against einops-order patches + float64 LayerNorm, row counts that end inside a pair of patches, NaN behind the last row."""
    C, p = 3, 16
    img = rnd(B, C, H, W, dtype=BF, seed=71) * 1.5 + 0.2
    P = C * p * p
    w = (1 + 0.1 * rnd(P, seed=72)).to(BF); b = (0.1 * rnd(P, seed=73)).to(BF) if bias else None
    h, ww = H // p, W // p
    rows = B * h * ww
    assert K.patch_ln_serves(img, C, H, W, p, p)
    ybuf = torch.full((rows + 2, P), float("nan"), dtype=BF, device=DEV)
    mbuf = torch.full((rows + 2,), float("nan"), device=DEV); rbuf = torch.full((rows + 2,), float("nan"), device=DEV)
    K.patch_ln_fwd(img, w, b, ybuf[:rows], mbuf[:rows], rbuf[:rows], B, C, H, W, p, p)
    patches = img.double().reshape(B, C, h, p, ww, p).permute(0, 2, 4, 3, 5, 1).reshape(rows, P)      # 'b c (h p1) (w p2) -> (b h w) (p1 p2 c)'
    ref = torch.nn.functional.layer_norm(patches, (P,), w.double(), b.double() if bias else None, 1e-5)
    assert rel(ybuf[:rows], ref) < 4e-3
    assert rel(mbuf[:rows], patches.mean(-1)) < 1e-5
    assert rel(rbuf[:rows], 1 / torch.sqrt(patches.var(-1, unbiased=False) + 1e-5)) < 1e-5
    assert torch.isnan(ybuf[rows:]).all() and torch.isnan(mbuf[rows:]).all()
    # the same bytes as patchify + the general LayerNorm kernel would give, to the rounding of the output
    pt = torch.empty(rows, P, dtype=BF, device=DEV); K.patchify(img, pt, B, C, H, W, p, p)
    assert torch.equal(pt.double(), patches)
    dy = rnd(rows, P, dtype=BF, seed=74)
    nblk = K.patch_ln_bwd_blocks(rows)
    part = torch.full((2 * nblk * P,), float("nan"), device=DEV)
    K.patch_ln_bwd_params(dy, img, mbuf[:rows], rbuf[:rows], part, B, C, H, W, p, p)
    dwv = torch.empty(P, device=DEV); dbv = torch.empty(P, device=DEV)
    K.colsum_partials(part, nblk, P, P, dwv); K.colsum_partials(part[nblk * P:], nblk, P, P, dbv)
    xh = (patches - patches.mean(-1, keepdim=True)) / torch.sqrt(patches.var(-1, unbiased=False, keepdim=True) + 1e-5)
    assert rel(dwv, (dy.double() * xh).sum(0)) < 1e-5
    assert rel(dbv, dy.double().sum(0)) < 1e-5


@pytest.mark.parametrize("M,N0,K0,N1,K1", [(50432, 2304, 768, 768, 768), (25216, 3072, 1024, 1024, 1024), (5000, 520, 264, 768, 256)])
@pytest.mark.parametrize("odt", [BF, F32])
def test_gemm_tn_pair(M, N0, K0, N1, K1, odt, monkeypatch):
    """vitk_gemm_tn_bf16_pair: two weight gradients over the same token rows in one launch of the four-wave kernel (shared split count,
    slabs [N0*K0 | N1*K1] per split) against float64, and bit-identical to the two single launches when those use the same split count."""
    splits = K.gemm_tn_pair_splits(M, N0, K0, N1, K1)
    assert splits > 0
    dY0 = rnd(M, N0, dtype=BF, seed=81) * (M ** -0.5); X0 = rnd(M, K0, dtype=BF, seed=82)
    dY1 = rnd(M, N1, dtype=BF, seed=83) * (M ** -0.5); X1 = rnd(M, K1, dtype=BF, seed=84)
    dW0 = torch.empty(N0, K0, dtype=odt, device=DEV); dW1 = torch.empty(N1, K1, dtype=odt, device=DEV)
    ws = torch.full((splits * (N0 * K0 + N1 * K1),), float("nan"), device=DEV)
    K.gemm_tn_bf16_pair(dY0, N0, X0, K0, dW0, dY1, N1, X1, K1, dW1, M, ws, splits)
    tol = 1e-5 if odt == F32 else 4e-3
    assert rel(dW0, dY0.double().t() @ X0.double()) < tol
    assert rel(dW1, dY1.double().t() @ X1.double()) < tol
    for dY, X, dW, n, k in ((dY0, X0, dW0, N0, K0), (dY1, X1, dW1, N1, K1)):
        ref = torch.empty(n, k, dtype=odt, device=DEV)
        w1 = torch.empty(splits * n * k, device=DEV)
        K.gemm_tn_bf16(dY, n, X, k, ref, k, M, n, k, w1, splits)           # same split count -> same partial sums, same fold order
        assert torch.equal(ref, dW)
    K.gemm_tn_bf16_pair(dY0, N0, X0, K0, dW0, dY1, N1, X1, K1, dW1, M, ws, splits, accumulate0=True)
    assert rel(dW0, 2 * (dY0.double().t() @ X0.double())) < 2 * tol
    assert rel(dW1, dY1.double().t() @ X1.double()) < tol


def test_fold_many_equals_the_one_job_folds_bit_for_bit():
    """vitk_fold_many (round 5: the LayerNorm finalizes and bias-gradient column sums of a layer's backward in one launch) against
    vitk_colsum_partials job by job: same additions in the same order, so the results are bit-identical -- float32 and 16-bit outputs,
    accumulate, a strided partial slab, more jobs than one launch holds (40), one-column and ragged widths."""
    g = torch.Generator(device="cpu"); g.manual_seed(5)
    jobs, want = [], []
    shapes = [(404, 3072, 3072), (394, 768, 768), (197, 768, 768), (1, 8, 8), (33, 100, 72), (16, 64, 64), (17, 1, 1)] * 7       # 49 jobs: two launches
    for i, (nparts, ld, cols) in enumerate(shapes):
        part = torch.randn(nparts * ld, generator=g).to(DEV)
        odt = F32 if i % 3 == 0 else BF
        acc = i % 4 == 1
        init = torch.randn(cols, generator=g).to(odt).to(DEV)
        a, b = init.clone(), init.clone()
        K.colsum_partials(part, nparts, ld, cols, a, acc)
        want.append(a)
        jobs.append((part, nparts, ld, cols, b, acc))
    K.fold_many(jobs)
    for i, (j, w) in enumerate(zip(jobs, want)):
        assert torch.equal(j[4], w), (i, shapes[i])
    # the three slabs of a LayerNorm backward's partials as three jobs == vitk_layernorm_bwd_finalize
    nblk, D = 200, 768
    partials = torch.randn(3 * nblk * D, generator=g).to(DEV)
    dw0, db0, dc0 = (torch.empty(D, dtype=BF, device=DEV) for _ in range(3))
    K.layernorm_bwd_finalize(partials, nblk, D, dw0, db0, dc0, K.dt(dw0))
    outs = [torch.empty(D, dtype=BF, device=DEV) for _ in range(3)]
    K.fold_many([(partials[i * nblk * D:], nblk, D, D, outs[i], False) for i in range(3)])
    assert torch.equal(outs[0], dw0) and torch.equal(outs[1], db0) and torch.equal(outs[2], dc0)
    with pytest.raises(L.VitkError):
        K.fold_many([(partials.to(BF), nblk, D, D, outs[0], False)])


def test_fused_backward_with_deferred_folds_equals_immediate_folds():
    """engine.TransformerFn.backward queues its folds (ops.deferred_folds) and flushes once per layer; with the queue switched off (every
    fold its own launch, as in rounds 1-4) the gradients must be bit-identical."""
    from vit_pytorch_amd import ViT, ops as OPS
    torch.manual_seed(0)
    m = ViT(image_size=64, patch_size=8, num_classes=10, dim=256, depth=3, heads=4, mlp_dim=512).to(DEV, dtype=BF)
    img = torch.randn(24, 3, 64, 64, device=DEV).to(BF)

    def grads():
        m.zero_grad(set_to_none=True)
        m(img).float().square().mean().backward()
        return [p.grad.clone() for p in m.parameters()]

    g1 = grads()
    real = OPS.deferred_folds
    class off:              # a deferred_folds that defers nothing
        def __enter__(self): return self
        def __exit__(self, *a): return False
    OPS.deferred_folds = off
    try:
        g0 = grads()
    finally:
        OPS.deferred_folds = real
    assert all(torch.equal(a, b) for a, b in zip(g1, g0))


@pytest.mark.parametrize("nparts", [1, 15, 16, 17, 127, 128, 129, 144, 394, 1000])
def test_fold_walk_sums_in_row_order_with_loads_in_flight(nparts):
    """layernorm.hip fold_walk (round 6: eight loads in flight per thread): out[c] = sum_p partials[p][c] is formed as 16 phase sums (rows ph, ph + 16, ...
    added in row order, float32) that are then added in phase order -- emulated here operation by operation, so the match is BIT-exact for every
    batch / tail split of the walk (full batches of 8, a tail of 0..7 rows, fewer rows than phases)."""
    cols = 200                                             # 3 full 64-column blocks + a partial one
    g = torch.Generator(device="cuda").manual_seed(nparts)
    part = torch.randn(nparts, cols, device="cuda", generator=g) * torch.logspace(-3, 3, cols, device="cuda")
    out = torch.empty(cols, device="cuda")
    K.colsum_partials(part, nparts, cols, cols, out)
    phase = torch.zeros(16, cols, device="cuda")
    for ph in range(16):
        for p in range(ph, nparts, 16):
            phase[ph] = phase[ph] + part[p]                # float32 adds, row order
    ref = torch.zeros(cols, device="cuda")
    for ph in range(16):
        ref = ref + phase[ph]
    assert torch.equal(out, ref), float((out - ref).abs().max())
    jobs = [(part, nparts, cols, cols, torch.empty(cols, device="cuda"), False)]
    K.fold_many(jobs)
    assert torch.equal(jobs[0][4], ref)
