"""GPU: NaViT path (BASELINE config 4): variable-length attention kernels against a per-image
PyTorch reference, and the drop-in NaViT module against the reference's golden outputs and the CPU oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import navit_oracle as NO  # noqa: E402
from oracle.params import NAVIT_BENCH_CASES, NAVIT_CASES, NAVIT_WIDE_CASES, make_navit_images, make_navit_params, navit_bench_sizes, sample_index  # noqa: E402
from vit_pytorch_amd import kernels as K  # noqa: E402
from vit_pytorch_amd.na_vit import NaViT, Segments  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    a = a.detach().double().flatten().cpu(); b = b.detach().double().flatten().cpu()
    n = b.norm().item()
    return (a - b).norm().item() / (n if n > 0 else 1.0)


@pytest.mark.parametrize("lens", [[197], [16, 300, 1, 129, 64], [1600, 7], [128, 128, 256]])
@pytest.mark.parametrize("H,d", [(1, 64), (3, 64), (2, 80), (2, 32), (3, 48), (2, 96)])
def test_varlen_attention_fwd_bwd(lens, H, d):
    I = H * d
    T = sum(lens)
    g = torch.Generator().manual_seed(len(lens) * 7 + H)
    q = (torch.randn(T, I, generator=g)).to(BF).to(DEV)
    kv = (torch.randn(T, 2 * I, generator=g)).to(BF).to(DEV)
    do = torch.randn(T, I, generator=g).to(BF).to(DEV)
    segs = Segments(lens, lens, torch.device(DEV))
    o = torch.empty(T, I, dtype=BF, device=DEV); lse = torch.empty(H, T, device=DEV)
    scale = 0.2
    K.attn_varlen_fwd_bf16(K.hnd(q, d, I), K.hnd(kv, d, 2 * I), K.hnd(kv, d, 2 * I, offset=I), K.hnd(o, d, I), lse, segs.cu_q,
                           segs.cu_k, segs.qblk_seg, segs.qblk_r0, segs.nqblk, T, H, d, scale)
    dq = torch.zeros_like(q); dkv = torch.zeros_like(kv); delta = torch.empty(H, T, device=DEV)
    K.attn_varlen_bwd_bf16(K.hnd(q, d, I), K.hnd(kv, d, 2 * I), K.hnd(kv, d, 2 * I, offset=I), K.hnd(o, d, I), K.hnd(do, d, I), lse, delta,
                           K.hnd(dq, d, I), K.hnd(dkv, d, 2 * I), K.hnd(dkv, d, 2 * I, offset=I), segs.cu_q, segs.cu_k, segs.qblk_seg,
                           segs.qblk_r0, segs.nqblk, segs.kblk_seg, segs.kblk_r0, segs.nkblk, T, H, d, scale)
    # reference: independent attention per image
    qd = q.double().requires_grad_(True); kvd = kv.double().requires_grad_(True)
    outs, start = [], 0
    for n in lens:
        qs = qd[start:start + n].view(n, H, d).transpose(0, 1)
        ks = kvd[start:start + n, :I].view(n, H, d).transpose(0, 1)
        vs = kvd[start:start + n, I:].view(n, H, d).transpose(0, 1)
        p = torch.softmax(qs @ ks.transpose(-1, -2) * scale, -1)
        outs.append((p @ vs).transpose(0, 1).reshape(n, I))
        start += n
    oref = torch.cat(outs)
    oref.backward(do.double())
    assert rel(o, oref) < 6e-3
    assert rel(dq, qd.grad) < 1.2e-2
    assert rel(dkv, kvd.grad) < 1.2e-2


def test_attention_pool_geometry_one_query_per_image():
    H, d = 2, 64
    I = H * d
    lens = [24, 4, 300]
    T = sum(lens); S = len(lens)
    g = torch.Generator().manual_seed(3)
    q = torch.randn(S, I, generator=g).to(BF).to(DEV); kv = torch.randn(T, 2 * I, generator=g).to(BF).to(DEV)
    segs = Segments([1] * S, lens, torch.device(DEV))
    o = torch.empty(S, I, dtype=BF, device=DEV); lse = torch.empty(H, S, device=DEV)
    K.attn_varlen_fwd_bf16(K.hnd(q, d, I), K.hnd(kv, d, 2 * I), K.hnd(kv, d, 2 * I, offset=I), K.hnd(o, d, I), lse, segs.cu_q,
                           segs.cu_k, segs.qblk_seg, segs.qblk_r0, segs.nqblk, S, H, d, 1.0)
    start = 0
    for s, n in enumerate(lens):
        qs = q[s].double().view(H, 1, d); ks = kv[start:start + n, :I].double().view(n, H, d).transpose(0, 1)
        vs = kv[start:start + n, I:].double().view(n, H, d).transpose(0, 1)
        ref = (torch.softmax(qs @ ks.transpose(-1, -2), -1) @ vs).reshape(I)
        assert rel(o[s], ref) < 6e-3
        start += n


@pytest.mark.parametrize("d", [64, 32, 80, 96, 128, 20, 256])
@pytest.mark.parametrize("dtype", [torch.float32, BF])
def test_rmsnorm_heads(dtype, d):
    """dim_head is free in the reference (na_vit.py:119): 16 lanes x 4 elements per 64-column chunk, d % 4 == 0, d <= 256"""
    T, H = 37, 3
    x = torch.randn(T, 2 * H * d).to(dtype).to(DEV)       # read the first half of a wider matrix (like to_kv's output)
    gamma = (1 + 0.2 * torch.randn(H, d)).to(dtype).to(DEV)
    y = torch.empty(T, H * d, dtype=dtype, device=DEV); rn = torch.empty(T * H, device=DEV)
    K.rmsnorm_heads_fwd(x, 2 * H * d, gamma, y, H * d, rn, T, H, d)
    xd = x[:, :H * d].double().view(T, H, d).requires_grad_(True); gd = gamma.double().requires_grad_(True)
    ref = torch.nn.functional.normalize(xd, dim=-1) * d ** 0.5 * gd
    tol = 3e-6 if dtype == torch.float32 else 5e-3
    assert rel(y, ref.reshape(T, H * d)) < tol
    dy = torch.randn(T, H * d).to(dtype).to(DEV)
    ref.backward(dy.double().view(T, H, d))
    dx = torch.zeros(T, 2 * H * d, dtype=dtype, device=DEV); dg = torch.empty(H, d, dtype=dtype, device=DEV)
    part = torch.empty(K.rmsnorm_heads_partials(T, H, d), device=DEV)
    K.rmsnorm_heads_bwd(dy, H * d, x, 2 * H * d, gamma, rn, dx, 2 * H * d, dg, part, T, H, d)
    assert rel(dx[:, :H * d], xd.grad.reshape(T, H * d)) < tol * 2
    assert rel(dg, gd.grad) < tol * 2
    assert dx[:, H * d:].abs().max().item() == 0


def _run_navit(case, dtype):
    params = make_navit_params(case["cfg"], case["seed"])
    imgs = make_navit_images(case["cfg"], case["sizes"], case["seed"] + 1000)
    m = NaViT(**case["cfg"])
    m.load_state_dict(params, strict=True)
    m = m.to(DEV, dtype=dtype).eval()
    out = m([[im.to(DEV, dtype=dtype) for im in g] for g in imgs])
    NO.O.loss_fn(out).backward()
    grads = {k: p.grad for k, p in m.named_parameters()}
    return out, grads, params, imgs


@pytest.mark.parametrize("name", list(NAVIT_CASES))
def test_navit_f32_matches_reference_golden(name):
    case = NAVIT_CASES[name]
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    out, grads, _, _ = _run_navit(case, torch.float32)
    assert tuple(out.shape) == gold["logits"].shape
    assert rel(out, torch.from_numpy(gold["logits"])) <= 1e-3
    for k, g in grads.items():
        assert rel(g, torch.from_numpy(gold["grad::" + k])) <= 1e-3, k


def _navit_wide_errors(name, dtype):
    """(logits error, gradient-sample error, the reference's own bf16 errors, worst per-tensor sample error in units of the tensor's
    share of its norm) of the drop-in at BASELINE config 4's width against the compact reference golden."""
    case = NAVIT_WIDE_CASES[name]
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    out, grads, _, _ = _run_navit(case, dtype)
    ref_logits = torch.from_numpy(gold["logits"])
    mine, ref, ref16 = [], [], []
    worst = 0.0
    for k, gr in grads.items():
        g = gr.detach().float().flatten().cpu()
        idx = torch.from_numpy(sample_index(g.numel(), case["sample"]))
        r = torch.from_numpy(gold["gsample::" + k]).float()
        mine.append(g[idx]); ref.append(r); ref16.append(torch.from_numpy(gold["bf16::gsample::" + k]).float())
        share = float(gold["gnorm::" + k]) * (r.numel() / g.numel()) ** 0.5
        worst = max(worst, (g[idx].double() - r.double()).norm().item() / max(share, 1e-30))
    cat = torch.cat
    return (rel(out, ref_logits), rel(cat(mine), cat(ref)), rel(torch.from_numpy(gold["bf16::logits"]), ref_logits),
            rel(cat(ref16), cat(ref)), worst)


@pytest.mark.parametrize("name", list(NAVIT_WIDE_CASES))
def test_navit_config4_width_f32_vs_reference_golden(name):
    """dim 1024, 16 heads, ONE pack of 4,096 tokens (32 images of five resolutions), depth 2: f32 mode pinned to the reference (1e-3)."""
    e, g, _, _, worst = _navit_wide_errors(name, torch.float32)
    print(f"{name} f32: logits {e:.2e} grad samples {g:.2e} worst tensor {worst:.2e}")
    assert e <= 1e-3 and g <= 1e-3 and worst <= 1e-2, (e, g, worst)


@pytest.mark.parametrize("name", list(NAVIT_WIDE_CASES))
def test_navit_config4_width_bf16_vs_reference_golden(name):
    """The packed-token MFMA path (fused packed stack, varlen attention, persistent GEMMs at M = 4,096) against the reference's f32
    outputs: held to 1.5x the error of the REFERENCE's own bf16 run on the same inputs (stored in the golden) + 1e-3."""
    e, g, e16, g16, worst = _navit_wide_errors(name, BF)
    print(f"{name} bf16: logits {e:.2e} (reference-bf16 {e16:.2e}) grad samples {g:.2e} (reference-bf16 {g16:.2e}) worst tensor {worst:.2e}")
    assert e <= 1.5 * e16 + 1e-3 and g <= 1.5 * g16 + 1e-3, (e, e16, g, g16)
    if NAVIT_WIDE_CASES[name]["cfg"]["depth"] <= 4:
        assert worst <= 0.15, worst
    else:
        # config 4's FULL depth (24 layers, round 6): bfloat16 gradients of this randomly initialised stack are poor in ANY bf16 pipeline --
        # [measured] the reference's own bf16 run is 5.1e-2 / 5.4e-1 (logits / gradient samples) from its float32 run, the drop-in
        # 3.8e-2 / 3.8e-1 -- so the per-tensor cap of the shallow cases does not apply; the drop-in must not be WORSE than the reference's bf16
        assert e <= e16 + 1e-3 and g <= g16 + 1e-3, (e, e16, g, g16)


@pytest.mark.parametrize("name", list(NAVIT_BENCH_CASES))
@pytest.mark.parametrize("dtype", [torch.float32, BF])
def test_navit_bench_workload_draw_vs_reference_golden(name, dtype):
    """BASELINE config 4's BENCH workload (bench.py --config navit: the 65-image draw of a*16 x b*16 px images, ~33 k tokens, grouped into
    nine packs of <= 4,096 tokens by the model itself) at config 4's width against outputs of /root/reference/vit_pytorch/na_vit.py on
    the same images and weights (oracle/make_golden.py::main_navit_bench; the depth the host's memory allows -- see oracle/params.py).
    f32: 1e-3; bf16 (what the bench runs): logits and gradient samples no further from the reference's f32 outputs than the reference's own
    bf16 run is (+ 1e-3) -- [measured] 1.7e-2 / 6.4e-2 against 2.6e-2 / 9.7e-2."""
    case = NAVIT_BENCH_CASES[name]
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    sizes = navit_bench_sizes()
    assert [tuple(x) for x in gold["sizes"].tolist()] == sizes
    params = make_navit_params(case["cfg"], case["seed"])
    imgs = make_navit_images(case["cfg"], [sizes], case["seed"] + 1000)[0]
    m = NaViT(**case["cfg"])
    m.load_state_dict(params, strict=True)
    m = m.to(DEV, dtype=dtype).eval()
    out = m([im.to(DEV, dtype=dtype) for im in imgs], group_images=True, group_max_seq_len=case["group_max_seq_len"])
    NO.O.loss_fn(out.float()).backward()
    ref_logits = torch.from_numpy(gold["logits"])
    mine, ref, ref16 = [], [], []
    for k, p in m.named_parameters():
        g = p.grad.detach().float().flatten().cpu()
        idx = torch.from_numpy(sample_index(g.numel(), case["sample"]))
        mine.append(g[idx]); ref.append(torch.from_numpy(gold["gsample::" + k]).float()); ref16.append(torch.from_numpy(gold["bf16::gsample::" + k]).float())
    e, g = rel(out, ref_logits), rel(torch.cat(mine), torch.cat(ref))
    e16, g16 = rel(torch.from_numpy(gold["bf16::logits"]), ref_logits), rel(torch.cat(ref16), torch.cat(ref))
    print(f"{name} {dtype}: logits {e:.2e} (reference-bf16 {e16:.2e}) grad samples {g:.2e} (reference-bf16 {g16:.2e})")
    if dtype == torch.float32:
        assert e <= 1e-3 and g <= 1e-3, (e, g)
    else:
        # twelve layers of dim 1024 on randn-scale weights: the REFERENCE's own bf16 run is 2.6e-2 / 9.7e-2 off its f32 run here, so the absolute
        # 2e-2 cap of the shallow cases cannot apply; the drop-in (f32 accumulation everywhere) must simply not be worse than the reference-bf16
        assert e <= 1.0 * e16 + 1e-3 and g <= 1.0 * g16 + 1e-3, (e, e16, g, g16)


@pytest.mark.parametrize("name", list(NAVIT_CASES))
def test_navit_bf16_vs_oracle(name):
    case = NAVIT_CASES[name]
    out, grads, params, imgs = _run_navit(case, BF)
    ref_out, ref_g = NO.run_fwd_bwd(case["cfg"], params, imgs, torch.float32)
    bf_out, bf_g = NO.run_fwd_bwd(case["cfg"], params, imgs, BF)
    keys = list(ref_g)
    cat = lambda d: torch.cat([d[k].detach().float().flatten().cpu() for k in keys])
    e, e_ref = rel(out, ref_out), rel(bf_out, ref_out)
    g, g_ref = rel(cat(grads), cat(ref_g)), rel(cat(bf_g), cat(ref_g))
    print(f"{name} bf16: logits {e:.2e} (reference-bf16 {e_ref:.2e}); grads {g:.2e} (reference-bf16 {g_ref:.2e})")
    assert e <= 1.5 * e_ref + 1e-3 and g <= 1.5 * g_ref + 1e-3


def test_navit_grouping_and_token_dropout_run():
    cfg = dict(NAVIT_CASES["navit_two_packs"]["cfg"], token_dropout_prob=0.25)
    m = NaViT(**cfg).to(DEV, dtype=BF)
    imgs = [torch.randn(3, h, w, device=DEV).to(BF) for (h, w) in [(32, 32), (64, 16), (8, 8), (48, 64), (24, 40)]]
    m.train()
    out = m(imgs, group_images=True, group_max_seq_len=48)
    assert out.shape == (5, cfg["num_classes"]) and torch.isfinite(out.float()).all()
    out.float().square().mean().backward()
    assert all(torch.isfinite(p.grad.float()).all() for p in m.parameters())
    m.eval()
    a = m(imgs); b = m([imgs[:2], imgs[2:]])            # packing must not change the result
    assert torch.equal(a, b)


def test_navit_fused_stack_vs_op_by_op():
    """The fused packed-token engine (engine.PackedTransformerFn) against the op-by-op module path and an f32 run."""
    cfg = dict(image_size=128, patch_size=8, num_classes=11, dim=256, depth=3, heads=4, mlp_dim=512)
    sizes = [[(64, 64), (128, 40), (8, 8)], [(96, 120), (16, 72)]]
    torch.manual_seed(3)
    base = NaViT(**cfg)
    sd = {k: v.clone() for k, v in base.state_dict().items()}
    g = torch.Generator().manual_seed(9)
    imgs = [[torch.randn(3, h, w, generator=g) for (h, w) in pack] for pack in sizes]

    def run(dtype, fused):
        m = NaViT(**cfg)
        m.load_state_dict(sd)
        m = m.to(DEV, dtype=dtype).eval()
        if not fused:
            m.transformer.norm.register_forward_hook(lambda *a: None)       # any hook turns the fused stack off
        assert m.transformer._fusable(torch.empty(1, dtype=dtype)) == (fused and dtype == BF)
        out = m([[im.to(DEV, dtype=dtype) for im in p] for p in imgs])
        NO.O.loss_fn(out).backward()
        return out, torch.cat([p.grad.detach().float().flatten() for p in m.parameters()])

    o32, g32 = run(torch.float32, False)
    of, gf = run(BF, True)
    ou, gu = run(BF, False)
    ef, eu = rel(of, o32), rel(ou, o32)
    df, du = rel(gf, g32), rel(gu, g32)
    print(f"fused: logits {ef:.2e} grads {df:.2e}; op-by-op: logits {eu:.2e} grads {du:.2e}")
    assert ef <= 1.5 * eu + 1e-3 and df <= 1.5 * du + 1e-3


@pytest.mark.parametrize("lens", [[197], [16, 300, 1, 129, 64]])
@pytest.mark.parametrize("H,d", [(3, 64), (2, 80)])
def test_varlen_attention_with_dropout(lens, H, d):
    """Attention dropout (na_vit.py:163 dropout_p) inside the packed kernels against a float64 reference with the same keep
    decisions: row = (head, packed query row), column = key index inside the image."""
    I = H * d
    T = sum(lens)
    p, seed, scale = 0.25, 9876, 0.2
    g = torch.Generator().manual_seed(len(lens) * 11 + H)
    q = torch.randn(T, I, generator=g).to(BF).to(DEV)
    kv = torch.randn(T, 2 * I, generator=g).to(BF).to(DEV)
    do = torch.randn(T, I, generator=g).to(BF).to(DEV)
    segs = Segments(lens, lens, torch.device(DEV))
    o = torch.empty(T, I, dtype=BF, device=DEV); lse = torch.empty(H, T, device=DEV)
    K.attn_varlen_fwd_bf16(K.hnd(q, d, I), K.hnd(kv, d, 2 * I), K.hnd(kv, d, 2 * I, offset=I), K.hnd(o, d, I), lse, segs.cu_q,
                           segs.cu_k, segs.qblk_seg, segs.qblk_r0, segs.nqblk, T, H, d, scale, p, seed)
    dq = torch.zeros_like(q); dkv = torch.zeros_like(kv); delta = torch.empty(H, T, device=DEV)
    K.attn_varlen_bwd_bf16(K.hnd(q, d, I), K.hnd(kv, d, 2 * I), K.hnd(kv, d, 2 * I, offset=I), K.hnd(o, d, I), K.hnd(do, d, I), lse, delta,
                           K.hnd(dq, d, I), K.hnd(dkv, d, 2 * I), K.hnd(dkv, d, 2 * I, offset=I), segs.cu_q, segs.cu_k, segs.qblk_seg,
                           segs.qblk_r0, segs.nqblk, segs.kblk_seg, segs.kblk_r0, segs.nkblk, T, H, d, scale, p, seed)
    keep = torch.empty(H * T, max(lens), dtype=torch.uint8, device=DEV)
    K.dropout_keep(keep, H * T, max(lens), p, seed)
    keep = keep.view(H, T, max(lens)).double() / (1 - p)
    qd = q.double().requires_grad_(True); kvd = kv.double().requires_grad_(True)
    outs, start = [], 0
    for n in lens:
        qs = qd[start:start + n].view(n, H, d).transpose(0, 1)
        ks = kvd[start:start + n, :I].view(n, H, d).transpose(0, 1)
        vs = kvd[start:start + n, I:].view(n, H, d).transpose(0, 1)
        pm = torch.softmax(qs @ ks.transpose(-1, -2) * scale, -1) * keep[:, start:start + n, :n]
        outs.append((pm @ vs).transpose(0, 1).reshape(n, I))
        start += n
    oref = torch.cat(outs)
    oref.backward(do.double())
    assert rel(o, oref) < 8e-3
    assert rel(dq, qd.grad) < 1.5e-2 and rel(dkv, kvd.grad) < 1.5e-2


def test_navit_trains_with_dropout():
    cfg = dict(NAVIT_CASES["navit_two_packs"]["cfg"], dropout=0.1, emb_dropout=0.1)
    torch.manual_seed(0)
    m = NaViT(**cfg).to(DEV, dtype=BF).train()
    imgs = [torch.randn(3, h, w, device=DEV).to(BF) for (h, w) in [(32, 32), (64, 16), (8, 8), (48, 64)]]
    out1 = m(imgs); out2 = m(imgs)
    assert torch.isfinite(out1.float()).all() and not torch.equal(out1, out2)       # fresh masks per call
    out1.float().square().mean().backward()
    assert all(torch.isfinite(q.grad.float()).all() for q in m.parameters())
    m.eval()
    assert torch.equal(m(imgs), m(imgs))


def test_packed_fused_dropout_layer_matches_masked_reference():
    """Active dropout (na_vit.py:101,103,163,172 at p = 0.1, training mode) runs inside the fused packed engine; its keep decisions
    are a counter hash, so a float64 reference that applies the very same masks (vitk_dropout_keep) must agree -- output and every
    gradient.  Sites of layer li: 4 li + {0: attention matrix, 1: after to_out, 2: after the GELU, 3: after the second Linear}."""
    from vit_pytorch_amd import engine as E
    from vit_pytorch_amd.na_vit import Transformer
    D, H, d, Fh, p = 256, 4, 64, 512, 0.1
    lens = [300, 512, 129, 83, 476]
    T, I = sum(lens), H * d
    torch.manual_seed(11)
    blk = Transformer(D, 1, H, d, Fh, dropout=p).to(DEV, dtype=BF).train()
    for q_ in blk.parameters():                                  # non-trivial gains / biases
        if q_.ndim == 1 or q_.shape[-2:] == (1, d):
            q_.data.add_(0.1 * torch.randn_like(q_))
    x = torch.randn(T, D, device=DEV).to(BF)
    segs = Segments(lens, lens, torch.device(DEV))
    assert blk._fusable(x) and blk._dropout_p() == p
    calls, salt = blk._drop_state()
    seed = (int(torch.initial_seed()) + 0x9E3779B1 * calls + 0x85EBCA6B * salt) & 0xffffffff
    y = blk(x, segs)
    assert blk._drop_state()[0] == calls + 1                    # the fused path drew its seeds
    NO.O.loss_fn(y).backward()

    def keep(rows, cols, k):
        m = torch.empty(rows, cols, dtype=torch.uint8, device=DEV)
        K.dropout_keep(m, rows, cols, p, E._hash32(seed + k))
        return m.double() / (1 - p)

    P = {k_: v.detach().double().requires_grad_(True) for k_, v in blk.named_parameters()}
    ln = lambda t, g_: torch.nn.functional.layer_norm(t, (D,), g_, None, 1e-5)
    rms = lambda t, g_: torch.nn.functional.normalize(t, dim=-1) * d ** 0.5 * g_.view(H, d)
    xd = x.double()
    a1 = ln(xd, P["layers.0.0.norm.gamma"])
    q = rms((a1 @ P["layers.0.0.to_q.weight"].t()).view(T, H, d), P["layers.0.0.q_norm.gamma"])
    kv = a1 @ P["layers.0.0.to_kv.weight"].t()
    k = rms(kv[:, :I].view(T, H, d), P["layers.0.0.k_norm.gamma"])
    v = kv[:, I:].view(T, H, d)
    km = keep(H * T, max(lens), 0).view(H, T, max(lens))
    outs, start = [], 0
    for n in lens:
        qs, ks, vs = (t[start:start + n].transpose(0, 1) for t in (q, k, v))
        pm = torch.softmax(qs @ ks.transpose(-1, -2), -1) * km[:, start:start + n, :n]      # scale 1: q and k are normalised
        outs.append((pm @ vs).transpose(0, 1).reshape(n, I))
        start += n
    o = torch.cat(outs)
    x2 = xd + (o @ P["layers.0.0.to_out.0.weight"].t()) * keep(T, D, 1)
    a2 = ln(x2, P["layers.0.1.0.gamma"])
    act = torch.nn.functional.gelu(a2 @ P["layers.0.1.1.weight"].t() + P["layers.0.1.1.bias"]) * keep(T, Fh, 2)
    x3 = x2 + (act @ P["layers.0.1.4.weight"].t() + P["layers.0.1.4.bias"]) * keep(T, D, 3)
    yref = ln(x3, P["norm.gamma"])
    NO.O.loss_fn(yref).backward()
    e = rel(y, yref)
    keys = list(P)
    g = rel(torch.cat([blk.get_parameter(k_).grad.float().flatten() for k_ in keys]), torch.cat([P[k_].grad.flatten() for k_ in keys]))
    worst = max(rel(blk.get_parameter(k_).grad, P[k_].grad) for k_ in keys)
    print(f"packed fused dropout layer: out {e:.2e}, grads {g:.2e}, worst tensor {worst:.2e}")
    assert e < 1e-2 and g < 2e-2 and worst < 6e-2
    blk.eval()
    assert blk._dropout_p() == 0.0 and blk._fusable(x)



@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 1e-1)])
def test_navit_image_gradient_matches_the_oracle(dtype, tol):
    """The reference is differentiable with respect to the input images through na_vit.py:300,350 (rearrange + to_patch_embedding); rounds
    1-5 raised instead.  `image.requires_grad_()` now returns d loss / d image of every image of every pack: against the CPU oracle's
    autograd on the same weights (float32: 1e-3 relative L2 per image, measured 3e-6; bfloat16: 1e-1, measured 5.7e-2 on the worst image -- the
    gradient passes two LayerNorms and every layer backwards in 16 bit), images without
    requires_grad get none, and the parameter gradients are unchanged by asking."""
    case = NAVIT_CASES["navit_two_packs"]
    params = make_navit_params(case["cfg"], case["seed"])
    imgs = make_navit_images(case["cfg"], case["sizes"], case["seed"] + 1000)
    m = NaViT(**case["cfg"])
    m.load_state_dict(params, strict=True)
    m = m.to(DEV, dtype=dtype).eval()
    mine = [[im.to(DEV, dtype=dtype).requires_grad_(True) for im in g] for g in imgs]
    mine[0][0].requires_grad_(False)                      # one image opts out
    out = m(mine)
    NO.O.loss_fn(out).backward()
    # the oracle on the CPU, float64 for a clean yardstick
    ref_imgs = [[im.double().requires_grad_(True) for im in g] for g in imgs]
    p64 = {k: v.detach().double().clone().requires_grad_(False) for k, v in params.items()}
    ref_out = NO.navit_fwd(ref_imgs, p64, patch_size=case["cfg"]["patch_size"], depth=case["cfg"]["depth"], heads=case["cfg"]["heads"])
    NO.O.loss_fn(ref_out).backward()
    assert mine[0][0].grad is None
    worst = 0.0
    for gi, (g_mine, g_ref) in enumerate(zip(mine, ref_imgs)):
        for ii, (a, b) in enumerate(zip(g_mine, g_ref)):
            if gi == 0 and ii == 0:
                continue
            assert a.grad is not None and a.grad.shape == a.shape and a.grad.dtype == dtype, (gi, ii)
            worst = max(worst, rel(a.grad, b.grad))
    print(f"NaViT image gradients {dtype}: worst per-image relative error {worst:.2e}")
    assert worst <= tol, worst


def test_navit_image_gradient_with_token_dropout_is_zero_on_dropped_patches():
    """Token dropout (na_vit.py:306-314) keeps a random subset of an image's patches: the gradient of a dropped patch's pixels is exactly zero,
    that of a kept patch is not."""
    torch.manual_seed(5)
    cfg = dict(image_size=64, patch_size=8, num_classes=7, dim=64, depth=1, heads=2, mlp_dim=64, token_dropout_prob=0.5)
    m = NaViT(**cfg).to(DEV).train()
    imgs = [[torch.randn(3, 32, 48, device=DEV).requires_grad_(True), torch.randn(3, 16, 16, device=DEV).requires_grad_(True)]]
    m(imgs).square().mean().backward()
    for im in imgs[0]:
        g = im.grad
        assert g is not None
        ph, pw = im.shape[1] // 8, im.shape[2] // 8
        per_patch = g.reshape(3, ph, 8, pw, 8).abs().amax(dim=(0, 2, 4))          # (ph, pw)
        kept = int((per_patch > 0).sum().item())
        assert kept == max(1, int(ph * pw * 0.5)), (kept, ph * pw)
