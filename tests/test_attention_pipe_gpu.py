"""GPU: the persistent, LDS-DMA-pipelined attention kernels (csrc/attention_pipe.hip) -- vit.py:55-63 and its autograd.

 * 16-bit flavour (ALL THREE kernels forced on: VITK_ATTN_PIPE=7; by default only the dQ kernel runs pipelined) against float64
   at every sequence length class the kernels serve (two to seven 32-row steps, ragged last steps, more items than resident
   workgroups so that buffers are recycled many times), bit-identical run to run, and against the one-workgroup-per-head
   kernels of attention.hip (VITK_ATTN_PIPE=0);
 * the f32-ACCURATE flavour (hi + lo operands, f32 outputs) against float64 to 1e-4: the "1e-3 logic gate" of the validation
   mode now covers the flash kernels themselves (staging, swizzle, masking, lazy maximum, both backward kernels)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from vit_pytorch_amd import kernels as K  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16


def rel(a, b):
    a = a.double().flatten(); b = b.double().flatten()
    n = b.norm().item()
    return (a - b).norm().item() / (n if n > 0 else 1.0)


def rnd(*shape, dtype=torch.float32, seed=0, scale=1.0):
    g = torch.Generator(device=DEV); g.manual_seed(seed)
    return (torch.randn(*shape, generator=g, device=DEV) * scale).to(dtype)


def attn_ref(qkv, do, H, d, scale):
    B, N, _ = qkv.shape
    I = H * d
    q, k, v = (qkv[..., i * I:(i + 1) * I].reshape(B, N, H, d).permute(0, 2, 1, 3).double() for i in range(3))
    q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
    s = (q @ k.transpose(-1, -2)) * scale
    o = torch.softmax(s, -1) @ v
    om = o.permute(0, 2, 1, 3).reshape(B, N, I)
    lse = torch.logsumexp(s, -1)
    om.backward(do.double())
    g = torch.cat([t.grad.permute(0, 2, 1, 3).reshape(B, N, I) for t in (q, k, v)], -1)
    return om.detach(), lse.detach(), g


def run16(qkv, do, H, d, scale, dtype=BF):
    B, N, _ = qkv.shape
    I = H * d
    o = torch.empty(B, N, I, dtype=dtype, device=DEV); lse = torch.empty(B, H, N, device=DEV)
    sb, sh, sn = N * 3 * I, d, 3 * I
    q_ = K.bhnd(qkv, sb, sh, sn); k_ = K.bhnd(qkv, sb, sh, sn, offset=I); v_ = K.bhnd(qkv, sb, sh, sn, offset=2 * I)
    o_ = K.bhnd(o, N * I, d, I)
    K.attn_fwd_bf16(q_, k_, v_, o_, lse, B, H, N, d, scale)
    dqkv = torch.full((B, N, 3 * I), float("nan"), dtype=dtype, device=DEV); delta = torch.empty(B, H, N, device=DEV)
    K.attn_bwd_bf16(q_, k_, v_, o_, K.bhnd(do, N * I, d, I), lse, delta, K.bhnd(dqkv, sb, sh, sn), K.bhnd(dqkv, sb, sh, sn, offset=I),
                    K.bhnd(dqkv, sb, sh, sn, offset=2 * I), B, H, N, d, scale)
    return o, lse, dqkv


# (B, H, N): N = 33 / 64 (two steps), 65, 100, 129 (ragged second half), 196 / 197 (ViT-B/L), 224 (full); B * H from fewer items than
# workgroups to 4x as many (every buffer recycled, the last round ragged)
SHAPES = [(2, 3, 197), (70, 12, 197), (3, 2, 33), (2, 2, 64), (5, 3, 65), (4, 4, 100), (3, 5, 129), (9, 12, 196), (2, 7, 224), (1, 1, 197),
          (171, 12, 50)]


@pytest.fixture(autouse=True)
def all_pipelined(monkeypatch):
    monkeypatch.setenv("VITK_ATTN_PIPE", "7")


@pytest.mark.parametrize("B,H,N", SHAPES)
@pytest.mark.parametrize("dtype", [BF, torch.float16])
def test_pipelined_attention_against_float64(B, H, N, dtype):
    d = 64
    scale = d ** -0.5
    qkv = rnd(B, N, 3 * H * d, dtype=dtype, seed=1, scale=1.5)
    do = rnd(B, N, H * d, dtype=dtype, seed=2)
    o, lse, dqkv = run16(qkv, do, H, d, scale, dtype)
    assert torch.isfinite(o.float()).all() and torch.isfinite(dqkv.float()).all()
    step = 16
    eo = eg = 0.0
    for b0 in range(0, B, step):
        oref, lref, gref = attn_ref(qkv[b0:b0 + step], do[b0:b0 + step], H, d, scale)
        eo = max(eo, rel(o[b0:b0 + step], oref)); eg = max(eg, rel(dqkv[b0:b0 + step], gref))
        assert (lse[b0:b0 + step].double() - lref).abs().max().item() < 4e-3
    tol_o, tol_g = (6e-3, 1.2e-2) if dtype == BF else (1e-3, 2e-3)
    assert eo < tol_o and eg < tol_g, (eo, eg)
    for _ in range(2):                                  # run-to-run bit identity: a race in the buffer hand-over shows up here
        o2, lse2, dqkv2 = run16(qkv, do, H, d, scale, dtype)
        assert torch.equal(o, o2) and torch.equal(lse, lse2) and torch.equal(dqkv, dqkv2)


@pytest.mark.parametrize("mask", [2, 1, 4])
def test_pipelined_equals_per_head_kernels(monkeypatch, mask):
    """Same inputs through attention.hip's kernels (VITK_ATTN_PIPE=0) and through each pipelined kernel alone (2 = the production
    default: dQ only): same fragment algebra, so the results agree to the last couple of bits of the 16-bit outputs (the split
    of the key range changes where the lazy maximum is re-based)."""
    B, H, N, d = 6, 12, 197, 64
    qkv = rnd(B, N, 3 * H * d, dtype=BF, seed=3, scale=1.5); do = rnd(B, N, H * d, dtype=BF, seed=4)
    monkeypatch.setenv("VITK_ATTN_PIPE", "0")
    o0, lse0, dqkv0 = run16(qkv, do, H, d, d ** -0.5)
    monkeypatch.setenv("VITK_ATTN_PIPE", str(mask))
    o, lse, dqkv = run16(qkv, do, H, d, d ** -0.5)
    assert rel(o, o0) < 2e-3 and rel(dqkv, dqkv0) < 4e-3
    assert (lse - lse0).abs().max().item() < 1e-4
    if mask == 2:
        I = H * d
        assert torch.equal(o, o0)                                       # the forward is the very same kernel
        assert torch.equal(dqkv[..., 2 * I:], dqkv0[..., 2 * I:])       # and so is dV (dK reads the delta the other dQ kernel summed in another order)


@pytest.mark.parametrize("B,H,N", [(3, 4, 197), (40, 12, 197), (2, 2, 65), (2, 3, 224), (3, 2, 100)])
def test_f32_accurate_flavour_against_float64(B, H, N):
    d = 64
    I = H * d
    scale = d ** -0.5
    qkv = rnd(B, N, 3 * I, seed=5, scale=1.5)           # f32 operands
    do = rnd(B, N, I, seed=6)
    hi = torch.empty(B, N, 3 * I, dtype=BF, device=DEV); lo = torch.empty_like(hi)
    K.split2(qkv, hi, lo)
    assert rel(hi.float() + lo.float(), qkv) < 2e-5
    sb, sh, sn = N * 3 * I, d, 3 * I
    pair = lambda off: (K.bhnd(hi, sb, sh, sn, offset=off), K.bhnd(lo, sb, sh, sn, offset=off))
    o = torch.empty(B, N, I, device=DEV); lse = torch.empty(B, H, N, device=DEV)
    o_ = K.bhnd(o, N * I, d, I)
    K.attn_fwd_x2(pair(0), pair(I), pair(2 * I), o_, lse, B, H, N, d, scale)
    dhi = torch.empty(B, N, I, dtype=BF, device=DEV); dlo = torch.empty_like(dhi)
    K.split2(do, dhi, dlo)
    dqkv = torch.full((B, N, 3 * I), float("nan"), device=DEV); delta = torch.empty(B, H, N, device=DEV)
    K.attn_bwd_x2(pair(0), pair(I), pair(2 * I), o_, (K.bhnd(dhi, N * I, d, I), K.bhnd(dlo, N * I, d, I)), lse, delta,
                  K.bhnd(dqkv, sb, sh, sn), K.bhnd(dqkv, sb, sh, sn, offset=I), K.bhnd(dqkv, sb, sh, sn, offset=2 * I), B, H, N, d, scale)
    eo = eg = el = 0.0
    for b0 in range(0, B, 16):
        oref, lref, gref = attn_ref(qkv[b0:b0 + 16], do[b0:b0 + 16], H, d, scale)
        eo = max(eo, rel(o[b0:b0 + 16], oref)); eg = max(eg, rel(dqkv[b0:b0 + 16], gref))
        el = max(el, (lse[b0:b0 + 16].double() - lref).abs().max().item())
    print(f"x2 flavour B={B} H={H} N={N}: o {eo:.2e} grads {eg:.2e} lse {el:.2e}")
    assert eo < 1e-4 and eg < 1e-4 and el < 1e-4, (eo, eg, el)


def test_rescale_branch_in_the_second_half():
    """One key in the LAST step dominates one query row: the lazy maximum is re-based after the buffer hand-over."""
    B, H, N, d = 1, 1, 197, 64
    scale = d ** -0.5
    qkv = rnd(B, N, 3 * d, dtype=BF, seed=7, scale=0.5)
    qkv[0, 5, :d] = 4.0
    qkv[0, 190, d:2 * d] = 4.0
    do = rnd(B, N, d, dtype=BF, seed=8)
    o, lse, dqkv = run16(qkv, do, H, d, scale)
    oref, lref, gref = attn_ref(qkv, do, H, d, scale)
    assert torch.isfinite(o.float()).all()
    assert rel(o, oref) < 6e-3 and rel(dqkv, gref) < 1.5e-2
    assert (lse.double() - lref).abs().max().item() < 2e-2


@pytest.mark.parametrize("B,H,N", [(2, 3, 197), (70, 12, 197), (9, 12, 196), (1, 1, 197), (3, 2, 208), (2, 2, 193), (300, 1, 197)])
@pytest.mark.parametrize("dtype", [BF, torch.float16])
def test_fused_backward_against_float64(B, H, N, dtype, monkeypatch):
    """The single-kernel backward (VITK_ATTN_PIPE bit 3: dQ, dK, dV in one pass over the scores, every operand through LDS-DMA rings)
    at the sequence lengths it serves (192 < N <= 208), from one item per workgroup to several, against float64; bit-identical run to
    run; the delta scratch output is written as the two-kernel path writes it."""
    monkeypatch.setenv("VITK_ATTN_PIPE", "8")
    d = 64
    scale = d ** -0.5
    qkv = rnd(B, N, 3 * H * d, dtype=dtype, seed=11, scale=1.5)
    do = rnd(B, N, H * d, dtype=dtype, seed=12)
    o, lse, dqkv = run16(qkv, do, H, d, scale, dtype)
    assert torch.isfinite(dqkv.float()).all()
    eg = 0.0
    for b0 in range(0, B, 16):
        oref, lref, gref = attn_ref(qkv[b0:b0 + 16], do[b0:b0 + 16], H, d, scale)
        eg = max(eg, rel(dqkv[b0:b0 + 16], gref))
        for name, sl in (("dq", slice(0, H * d)), ("dk", slice(H * d, 2 * H * d)), ("dv", slice(2 * H * d, 3 * H * d))):
            r = rel(dqkv[b0:b0 + 16, :, sl], gref[..., sl])
            assert r < (1.2e-2 if dtype == BF else 2e-3), (name, r)
    for _ in range(2):
        o2, lse2, dqkv2 = run16(qkv, do, H, d, scale, dtype)
        assert torch.equal(dqkv, dqkv2)
    monkeypatch.setenv("VITK_ATTN_PIPE", "0")
    o0, lse0, dqkv0 = run16(qkv, do, H, d, scale, dtype)
    assert rel(dqkv, dqkv0) < 4e-3
