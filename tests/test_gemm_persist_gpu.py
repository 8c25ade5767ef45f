"""GPU: the persistent NT GEMM (csrc/gemm_nt_persist.hip) against float64 products of the same operands.

Covers what the kernel adds over the per-tile one: the continuous K-step stream across tiles (several tiles per
workgroup), both tile heights in one launch, partial last rows / columns, very short K (one and two K-steps per tile, so
the DMA prologue spans tiles), every epilogue, the in-register bias-gradient sums, and run-to-run bit identity (a race in
the LDS ring or in the counted waits shows up as a changing result)."""
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


# (M, N, K): many tiles per workgroup + tail (ViT-B quarter batch), partial rows and columns, K of one / two / three K-steps
SHAPES = [(12608, 768, 768), (12608, 3072, 768), (12608, 768, 3072), (70001, 520, 96), (9000, 1000, 32), (33000, 264, 64), (1024, 256, 64),
          (4100, 2304, 128)]


def _all_epilogues(M, N, Kd, A, W, bias, resid, h):
    out = {}
    C = torch.empty(M, N, dtype=BF, device=DEV)
    K.gemm_nt_bf16(A, Kd, W, Kd, C, N, M, N, Kd)
    out["none"] = C.clone()
    K.gemm_nt_bf16(A, Kd, W, Kd, C, N, M, N, Kd, L.EPI_BIAS, bias=bias)
    out["bias"] = C.clone()
    aux = torch.empty(M, N, dtype=BF, device=DEV)
    K.gemm_nt_bf16(A, Kd, W, Kd, C, N, M, N, Kd, L.EPI_BIAS_GELU, bias=bias, aux=aux)
    out["gelu"] = C.clone(); out["pre"] = aux
    o32 = torch.empty(M, N, device=DEV)
    K.gemm_nt_bf16(A, Kd, W, Kd, o32, N, M, N, Kd, L.EPI_RESID, bias=bias, resid=resid)
    out["resid"] = o32
    R = K.gemm_nt_colsum_rows(M, N, Kd, N)
    part = torch.full((R * N,), float("nan"), device=DEV)
    C2 = torch.empty(M, N, dtype=BF, device=DEV)
    K.gemm_nt_bf16_gelu_bwd_colsum(A, Kd, W, Kd, C2, N, M, N, Kd, h, part)
    out["gbwd"] = C2; out["part"] = part.view(R, N)
    return out


@pytest.mark.parametrize("M,N,Kd", SHAPES)
def test_persistent_nt_against_float64(M, N, Kd):
    plan = K.gemm_nt_plan(M, N, Kd, N)
    assert plan["persistent"]
    A = rnd(M, Kd, dtype=BF, seed=1); W = (rnd(N, Kd, seed=2) * Kd ** -0.5).to(BF)
    bias = rnd(N, dtype=BF, seed=3); resid = rnd(M, N, seed=4); h = rnd(M, N, dtype=BF, seed=5)
    ref = A.double() @ W.double().t()
    got = _all_epilogues(M, N, Kd, A, W, bias, resid, h)
    assert rel(got["none"], ref) < 4e-3
    # exact up to the output rounding: equal to the bf16 rounding of the f32-accumulated product within 2 ulp of the largest value
    assert (got["none"].double() - ref.float().to(BF).double()).abs().max().item() <= 2 * 2 ** -8 * ref.abs().max().item()
    pre = ref + bias.double()
    assert rel(got["bias"], pre) < 4e-3
    assert rel(got["pre"], pre) < 4e-3 and rel(got["gelu"], torch.nn.functional.gelu(pre)) < 4e-3
    assert rel(got["resid"], resid.double() + pre) < 1e-5
    hd = h.double().requires_grad_(True)
    torch.nn.functional.gelu(hd).backward(ref)
    assert rel(got["gbwd"], hd.grad) < 4e-3
    assert not torch.isnan(got["part"]).any()
    assert rel(got["part"].double().sum(0), got["gbwd"].double().sum(0)) < 1e-5
    # run-to-run bit identity
    for _ in range(2):
        again = _all_epilogues(M, N, Kd, A, W, bias, resid, h)
        for k in got:
            assert torch.equal(got[k], again[k]), k


def test_plan_uses_both_tile_heights_at_vit_b_sizes():
    p = K.gemm_nt_plan(50432, 768, 3072, 768)
    assert p["persistent"] and p["tiles_m256"] > 0 and p["tiles_m128"] > 0
    assert 256 * p["tiles_m256"] + 128 * p["tiles_m128"] >= 50432
    os.environ["VITK_NT_W128"] = "0"          # the 8-wave kernel alone (round 5: the four-wave kernel takes the whole rounds, tests/test_gemm_nt_w128_gpu.py)
    try:
        assert K.gemm_nt_colsum_rows(50432, 768, 3072, 768) == 2 * (p["tiles_m256"] + p["tiles_m128"])
    finally:
        os.environ.pop("VITK_NT_W128", None)
    assert not K.gemm_nt_plan(512, 768, 768, 768)["persistent"]          # small M stays on the 128-row kernel


def test_persistent_nt_in_place_residual_and_strided_operands():
    """The engine adds into the residual stream in place and reads q|k|v slices of the merged projection by leading dimension."""
    M, N, Kd, ld = 5000, 768, 256, 1024
    Abig = rnd(M, ld, dtype=BF, seed=11); W = (rnd(N, Kd, seed=12) * Kd ** -0.5).to(BF)
    A = Abig[:, 256:512]
    ref = A.double() @ W.double().t()
    x = rnd(M, N, seed=13); x0 = x.clone()
    L.check(L.load().vitk_gemm_nt_bf16(A.data_ptr(), ld, W.data_ptr(), Kd, x.data_ptr(), N, M, N, Kd, L.EPI_RESID, None, x.data_ptr(), None,
                                       torch.cuda.current_stream().cuda_stream), "gemm_nt_bf16")
    assert rel(x, x0.double() + ref) < 1e-5


@pytest.mark.parametrize("M,N,Kd", [(12608, 768, 768), (4100, 2304, 128), (9000, 1000, 32), (70001, 520, 96), (12608, 768, 3072)])
def test_k_blocked_weight_gives_identical_results(M, N, Kd):
    """vitk_pack_w_nt: the K-blocked copy of W (ldw = 0) feeds the same arithmetic -- every epilogue bit-identical to the
    row-major operand; the transposed flavour equals packing an explicit transpose."""
    A = rnd(M, Kd, dtype=BF, seed=21); W = (rnd(N, Kd, seed=22) * Kd ** -0.5).to(BF)
    bias = rnd(N, dtype=BF, seed=23); resid = rnd(M, N, seed=24); h = rnd(M, N, dtype=BF, seed=25)
    pf = torch.empty(K.pack_w_nt_bytes(N, Kd) // 2, dtype=BF, device=DEV)
    pt = torch.empty(K.pack_w_nt_bytes(Kd, N) // 2, dtype=BF, device=DEV) if N % 32 == 0 else None
    K.pack_w_nt(W, Kd, N, Kd, pf, pt)
    ref = _all_epilogues(M, N, Kd, A, W, bias, resid, h)

    def packed_epilogues(P):
        out = {}
        C = torch.empty(M, N, dtype=BF, device=DEV)
        K.gemm_nt_bf16(A, Kd, P, 0, C, N, M, N, Kd)
        out["none"] = C.clone()
        K.gemm_nt_bf16(A, Kd, P, 0, C, N, M, N, Kd, L.EPI_BIAS, bias=bias)
        out["bias"] = C.clone()
        aux = torch.empty(M, N, dtype=BF, device=DEV)
        K.gemm_nt_bf16(A, Kd, P, 0, C, N, M, N, Kd, L.EPI_BIAS_GELU, bias=bias, aux=aux)
        out["gelu"] = C.clone(); out["pre"] = aux
        o32 = torch.empty(M, N, device=DEV)
        K.gemm_nt_bf16(A, Kd, P, 0, o32, N, M, N, Kd, L.EPI_RESID, bias=bias, resid=resid)
        out["resid"] = o32
        R = K.gemm_nt_colsum_rows(M, N, Kd, N)
        part = torch.full((R * N,), float("nan"), device=DEV)
        C2 = torch.empty(M, N, dtype=BF, device=DEV)
        K.gemm_nt_bf16_gelu_bwd_colsum(A, Kd, P, 0, C2, N, M, N, Kd, h, part)
        out["gbwd"] = C2; out["part"] = part.view(R, N)
        return out

    got = packed_epilogues(pf)
    for k in ref:
        assert torch.equal(ref[k], got[k]), k
    if pt is not None and K.gemm_nt_plan(M, Kd, N, Kd)["persistent"]:
        # dX = dY . W  ==  NT GEMM with the operand W^T (rows Kd, reduction N)
        dY = rnd(M, N, dtype=BF, seed=26)
        Wt = W.t().contiguous()
        d1 = torch.empty(M, Kd, dtype=BF, device=DEV); d2 = torch.empty(M, Kd, dtype=BF, device=DEV)
        K.gemm_nt_bf16(dY, N, Wt, N, d1, Kd, M, Kd, N)
        K.gemm_nt_bf16(dY, N, pt, 0, d2, Kd, M, Kd, N)
        assert torch.equal(d1, d2)
        pt2 = torch.empty_like(pt)
        K.pack_w_nt(Wt, N, Kd, N, pt2, None)
        assert torch.equal(pt, pt2)


def test_k_blocked_weight_refused_where_the_persistent_kernel_does_not_run():
    M, N, Kd = 512, 768, 768
    A = rnd(M, Kd, dtype=BF, seed=31); W = rnd(N, Kd, dtype=BF, seed=32)
    pf = torch.empty(K.pack_w_nt_bytes(N, Kd) // 2, dtype=BF, device=DEV)
    K.pack_w_nt(W, Kd, N, Kd, pf, None)
    C = torch.empty(M, N, dtype=BF, device=DEV)
    with pytest.raises(L.VitkError):
        K.gemm_nt_bf16(A, Kd, pf, 0, C, N, M, N, Kd)


@pytest.mark.parametrize("M,N,Kd", [(12608, 768, 768), (12608, 3072, 768), (12608, 768, 3072), (33000, 520, 256)])
def test_dynamic_tickets_equal_static_lists(M, N, Kd, monkeypatch):
    """Dynamic tile tickets (K >= 256; on while vitk_set_cu_reserve > 0 or with VITK_NTP_DYNAMIC=1) and the static lists compute every tile the same way: all
    epilogues bit-identical -- also while another kernel holds 40 CUs (vitk_test_occupy_cus), when the late workgroups of a
    dynamic launch find the queues dry and the resident ones draw the other XCDs' tiles."""
    A = rnd(M, Kd, dtype=BF, seed=41); W = (rnd(N, Kd, seed=42) * Kd ** -0.5).to(BF)
    bias = rnd(N, dtype=BF, seed=43); resid = rnd(M, N, seed=44); h = rnd(M, N, dtype=BF, seed=45)
    monkeypatch.setenv("VITK_NTP_STATIC", "1")
    ref = _all_epilogues(M, N, Kd, A, W, bias, resid, h)
    monkeypatch.delenv("VITK_NTP_STATIC")
    monkeypatch.setenv("VITK_NTP_DYNAMIC", "1")
    got = _all_epilogues(M, N, Kd, A, W, bias, resid, h)
    for k in ref:
        assert torch.equal(ref[k], got[k]), k
    hog = torch.cuda.Stream(priority=-1)
    for _ in range(2):
        L.check(L.load().vitk_test_occupy_cus(40, 3.0, hog.cuda_stream), "occupy")
        torch.cuda._sleep(100000)
        got = _all_epilogues(M, N, Kd, A, W, bias, resid, h)
        torch.cuda.synchronize()
        for k in ref:
            if k != "part":                     # (the column-sum partial rows are per m-tile: same values, checked via their sum)
                assert torch.equal(ref[k], got[k]), k
        assert rel(got["part"].double().sum(0), ref["part"].double().sum(0)) < 1e-6


@pytest.mark.parametrize("M,N,Kd", [(50432, 3072, 768), (1182, 3072, 768), (2049, 512, 256), (4616, 5120, 1280)])
def test_gelu_factor_epilogue_pair(M, N, Kd):
    """EPI_BIAS_GELU_DG (forward: C = gelu(pre), aux = gelu'(pre) of the rounded pre-activation) and EPI_MUL_AUX (backward: C = acc * aux +
    column sums) of the persistent kernel against float64 -- and against the pair they replace (EPI_BIAS_GELU saving pre + EPI_GELU_BWD):
    the same gelu output bit for bit, the backward product within one more 16-bit rounding."""
    import math
    from vit_pytorch_amd import ops
    if not ops.gelu_dg_ok(BF, M, N, Kd):
        pytest.skip("shape not served")
    torch.manual_seed(5)
    A = (torch.randn(M, Kd, device=DEV) * 0.7).to(BF); W = (torch.randn(N, Kd, device=DEV) * Kd ** -0.5).to(BF); b = (torch.randn(N, device=DEV) * 0.3).to(BF)
    act, dg = ops.linear_fwd(A, W, b, M, gelu=True, save_dg=True)
    act0, pre0 = ops.linear_fwd(A, W, b, M, gelu=True)
    assert torch.equal(act, act0)
    pre = (A.double() @ W.double().t() + b.double()).to(BF).double()        # the kernel rounds the pre-activation first
    phi = 0.5 * (1 + torch.erf(pre / math.sqrt(2)))
    dref = phi + pre * torch.exp(-pre * pre / 2) / math.sqrt(2 * math.pi)
    # the pre-activation itself carries the f32-accumulation error of one rounding step on a few elements: compare where both agree
    same = pre0.double() == pre
    assert same.double().mean().item() > 0.99
    # the stored factor: 8-bit fixed-point codes (the default: |error| <= 0.0025 over the whole range) or the 16-bit type (VITK_GELU_DG=16)
    if dg.dtype == torch.uint8:
        dgv = (dg.double() - 27.0) * 0.005
        assert (dgv - dref)[same].abs().max().item() <= 0.0025 + 1e-4
    else:
        dgv = dg.double()
        assert (dgv - dref)[same].abs().max().item() <= 2 ** -8 * 1.2 + 1e-4
    # backward: dY (M, N2) . W2 (N2, N) * factor, N2 = Kd
    dY = (torch.randn(M, Kd, device=DEV) * 0.5).to(BF); W2 = (torch.randn(Kd, N, device=DEV) * Kd ** -0.5).to(BF)
    db = torch.empty(N, dtype=BF, device=DEV); db0 = torch.empty(N, dtype=BF, device=DEV)
    dx, done = ops.linear_dx(dY, W2, M, gelu_dg=dg, db=db)
    dx0, done0 = ops.linear_dx(dY, W2, M, gelu_pre=pre0, db=db0)
    assert done and done0
    ref = (dY.double() @ W2.double()) * dgv
    assert rel(dx, ref) < 4e-3
    assert rel(db, dx.double().sum(0)) < 4e-3              # column sums of the ROUNDED output, like GELU_BWD
    assert rel(dx, dx0.double()) < 6e-3                     # vs the pair it replaces: one more rounding of the factor


def test_pack_w_nt_many_equals_the_single_packs():
    """vitk_pack_w_nt_many (round 6): a table of weights per launch -- 13 layers' worth (104 weights, 195 packs: more than one 96-job table) of
    mixed shapes, some with one side only -- byte for byte what vitk_pack_w_nt writes for each."""
    shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072), (1000, 96), (520, 32), (96, 1024), (264, 64)] * 13
    rows, refs = [], []
    for i, (N, Kd) in enumerate(shapes):
        W = (rnd(N, Kd, seed=300 + i) * Kd ** -0.5).to(BF)
        want_f = Kd % 32 == 0 and i % 5 != 4
        want_t = N % 32 == 0 and i % 7 != 6
        if not (want_f or want_t):
            want_f = True
        pf = torch.full((K.pack_w_nt_bytes(N, Kd) // 2,), -7.0, dtype=BF, device=DEV) if want_f else None
        pt = torch.full((K.pack_w_nt_bytes(Kd, N) // 2,), -7.0, dtype=BF, device=DEV) if want_t else None
        rf = torch.empty_like(pf) if want_f else None
        rt = torch.empty_like(pt) if want_t else None
        K.pack_w_nt(W, Kd, N, Kd, rf, rt)
        rows.append((W, N, Kd, pf, pt)); refs.append((rf, rt))
    assert sum((r[3] is not None) + (r[4] is not None) for r in rows) > 96
    K.pack_w_nt_many(rows)
    for (W, N, Kd, pf, pt), (rf, rt) in zip(rows, refs):
        if pf is not None:
            assert torch.equal(pf.view(torch.int16), rf.view(torch.int16)), (N, Kd, "forward pack")
        if pt is not None:
            assert torch.equal(pt.view(torch.int16), rt.view(torch.int16)), (N, Kd, "transposed pack")
    K.pack_w_nt_many([])                                   # an empty table is nothing
    with pytest.raises(L.VitkError):
        K.pack_w_nt_many([(rows[4][0], 1000, 96, None, torch.empty(8, dtype=BF, device=DEV))])      # N = 1000: no transposed pack (N % 32)
