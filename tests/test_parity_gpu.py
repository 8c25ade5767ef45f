"""GPU parity of the drop-in modules against (a) the golden vectors produced by the reference
itself (tests/golden, oracle/make_golden.py) and (b) the CPU oracle run live on the same
deterministic weights/inputs.

Tolerances (floating point, stated here as the task requires):
  * f32 mode (params float32; same host logic, f32 kernels): logits and every parameter gradient
    within 1e-3 relative L2 of the reference (north_star's bound); measured ~1e-6..1e-5.
  * bf16 mode (production): the reference's OWN bf16 run deviates from its f32 run by 3.6e-3
    (autocast) .. 9.3e-3 (pure bf16) on ViT-B/16 logits, so 1e-3 is not a
    physical bound for any bf16 pipeline.  Gate: error vs the f32 oracle <= 1.5x the error of the
    oracle's own pure-bf16 CPU run on the same inputs (+1e-3 absolute slack), for logits and for
    the concatenated gradient vector.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import vit_oracle as O  # noqa: E402
from oracle.params import CASES, WIDE_CASES, make_images, make_params, sample_index  # noqa: E402
from vit_pytorch_amd import SimpleViT, ViT  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda"


def rel(a, b):
    a = a.detach().double().flatten().cpu(); b = b.detach().double().flatten().cpu()
    n = b.norm().item()
    return (a - b).norm().item() / (n if n > 0 else 1.0)


def build(kind, cfg, params, dtype):
    cls = ViT if kind == "vit" else SimpleViT
    m = cls(**cfg)
    m.load_state_dict(params, strict=True)
    return m.to(DEV, dtype=dtype)


def run_mine(kind, cfg, params, img, dtype, loss_scale=1.0):
    """loss_scale: IEEE half has 5 exponent bits -- the 1e-6-sized gradients of a mean-of-squares loss behind a mean
    pool fall into its subnormals -- so fp16 runs scale the loss (as any fp16 training does) and unscale the gradients."""
    m = build(kind, cfg, params, dtype)
    out = m(img.to(DEV, dtype=dtype))
    (O.loss_fn(out) * loss_scale).backward()
    grads = {k: ((p.grad.float() / loss_scale) if p.grad is not None else torch.zeros_like(p)) for k, p in m.named_parameters()}
    return out, grads


@pytest.mark.parametrize("name", list(CASES))
def test_f32_mode_matches_reference_golden(name):
    case = CASES[name]
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    params = make_params(case["kind"], case["cfg"], case["seed"])
    img = make_images(case["cfg"], case["batch"], case["seed"] + 1000, case.get("image"))
    out, grads = run_mine(case["kind"], case["cfg"], params, img, torch.float32)
    assert tuple(out.shape) == gold["logits"].shape
    e = rel(out, torch.from_numpy(gold["logits"]))
    assert e <= 1e-3, e
    assert e <= 5e-5, f"f32 mode should be round-off close, got {e}"
    for k in params:
        g_ref = torch.from_numpy(gold["grad::" + k])
        if g_ref.numel() == 0:
            assert grads[k].numel() == 0
            continue
        eg = rel(grads[k], g_ref)
        assert eg <= 1e-3, (k, eg)


@pytest.mark.parametrize("name", list(CASES))
def test_fp16_mode_vs_golden(name):
    """model.half() (libvitk_f16.so: the same kernels with IEEE-half operands, v_mfma_..._f16, f32 accumulation).
    Half carries 11 significant bits against bfloat16's 8, so the gate is an absolute one: 3e-3 relative L2 of the
    reference's golden logits / gradients (measured ~5e-4)."""
    case = CASES[name]
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    params = make_params(case["kind"], case["cfg"], case["seed"])
    img = make_images(case["cfg"], case["batch"], case["seed"] + 1000, case.get("image"))
    out, grads = run_mine(case["kind"], case["cfg"], params, img, torch.float16, loss_scale=256.0)
    assert out.dtype == torch.float16
    keys = [k for k in params if params[k].numel() > 0]
    cat = lambda get: torch.cat([get(k).detach().float().flatten().cpu() for k in keys])
    e = rel(out, torch.from_numpy(gold["logits"]))
    g = rel(cat(lambda k: grads[k]), cat(lambda k: torch.from_numpy(gold["grad::" + k])))
    print(f"{name} fp16: logits {e:.2e} grads {g:.2e}")
    assert e <= 3e-3 and g <= 3e-3


@pytest.mark.parametrize("name", list(CASES))
def test_bf16_mode_vs_oracle(name):
    case = CASES[name]
    params = make_params(case["kind"], case["cfg"], case["seed"])
    img = make_images(case["cfg"], case["batch"], case["seed"] + 1000, case.get("image"))
    ref_out, ref_g = O.run_fwd_bwd(case["kind"], case["cfg"], params, img, torch.float32)
    bf_out, bf_g = O.run_fwd_bwd(case["kind"], case["cfg"], params, img, torch.bfloat16)   # the reference's own bf16 behaviour
    out, grads = run_mine(case["kind"], case["cfg"], params, img, torch.bfloat16)
    keys = [k for k in params if params[k].numel()]
    cat = lambda d: torch.cat([d[k].detach().float().flatten().cpu() for k in keys])
    e_mine, e_ref = rel(out, ref_out), rel(bf_out, ref_out)
    g_mine, g_ref = rel(cat(grads), cat(ref_g)), rel(cat(bf_g), cat(ref_g))
    print(f"{name}: logits err mine {e_mine:.2e} vs reference-bf16 {e_ref:.2e}; grads mine {g_mine:.2e} vs {g_ref:.2e}")
    assert e_mine <= 1.5 * e_ref + 1e-3, (e_mine, e_ref)
    assert g_mine <= 1.5 * g_ref + 1e-3, (g_mine, g_ref)


def _wide_errors(name, dtype, loss_scale=1.0):
    """(logits error, gradient-sample error, reference-bf16 logits error, reference-bf16 gradient-sample error, worst per-tensor
    sample error in units of that tensor's share of its norm) of the drop-in at one of WIDE_CASES against the compact golden."""
    case = WIDE_CASES[name]
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    params = make_params(case["kind"], case["cfg"], case["seed"])
    img = make_images(case["cfg"], case["batch"], case["seed"] + 1000)
    out, grads = run_mine(case["kind"], case["cfg"], params, img, dtype, loss_scale=loss_scale)
    ref_logits = torch.from_numpy(gold["logits"])
    keys = [k for k in params if params[k].numel()]
    mine, ref, ref16 = [], [], []
    worst = 0.0
    for k in keys:
        g = grads[k].detach().float().flatten().cpu()
        idx = torch.from_numpy(sample_index(g.numel(), case.get("sample", 4096)))
        r = torch.from_numpy(gold["gsample::" + k]).float()
        mine.append(g[idx]); ref.append(r); ref16.append(torch.from_numpy(gold["bf16::gsample::" + k]).float())
        share = float(gold["gnorm::" + k]) * (r.numel() / g.numel()) ** 0.5        # expected norm of the sample
        worst = max(worst, (g[idx].double() - r.double()).norm().item() / max(share, 1e-30))
    cat = torch.cat
    return (rel(out, ref_logits), rel(cat(mine), cat(ref)), rel(torch.from_numpy(gold["bf16::logits"]), ref_logits),
            rel(cat(ref16), cat(ref)), worst)


@pytest.mark.parametrize("name", list(WIDE_CASES))
def test_production_widths_f32_vs_reference_golden(name):
    """f32 mode at BASELINE config 2 / 3 / 5 layer shapes, pinned DIRECTLY to outputs of the reference (north star: 1e-3)."""
    e, g, _, _, worst = _wide_errors(name, torch.float32)
    print(f"{name} f32: logits {e:.2e} grad samples {g:.2e} worst tensor {worst:.2e}")
    assert e <= 1e-3 and g <= 1e-3 and worst <= 1e-2, (e, g, worst)


@pytest.mark.parametrize("name", list(WIDE_CASES))
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_production_widths_16bit_vs_reference_golden(name, dtype):
    """The MFMA path (persistent NT GEMM: M = batch * N >= 1024; whole-head / chunked attention; split-M weight-gradient GEMM) at
    BASELINE config 2 / 3 / 5 layer shapes against the reference's own f32 outputs.  bf16 is held to 1.5x the error of the
    REFERENCE's pure-bf16 run on the same inputs (stored in the golden) + 1e-3; fp16 (11-bit significand) to 3e-3 outright."""
    fp16 = dtype == torch.float16
    e, g, e16, g16, worst = _wide_errors(name, dtype, loss_scale=4096.0 if fp16 else 1.0)
    print(f"{name} {dtype}: logits {e:.2e} (reference-bf16 {e16:.2e}) grad samples {g:.2e} (reference-bf16 {g16:.2e}) worst tensor {worst:.2e}")
    if fp16:
        assert e <= 3e-3 and g <= 3e-3, (e, g)
    else:
        assert e <= 1.5 * e16 + 1e-3 and g <= 1.5 * g16 + 1e-3, (e, e16, g, g16)
        # at large batch the reference's OWN bf16 gradients are poor (its bf16 sums over 50,432 token rows: 2.0e-1 at batch 256), which
        # would make the relative rule empty there: the drop-in (f32 accumulation everywhere) is also held to an absolute 2e-2
        assert g <= 2e-2, (g, g16)
    assert worst <= 0.15, worst


def test_recompute_path_vs_reference_golden(monkeypatch):
    """BASELINE config 5's architecture (ViT-H/14 at 336 px, dim_head 80) at depth 4 / batch 8 with the activation-recompute policy
    FORCED ON (engine._recompute_policy: what config 5 takes at batch 256 / GPU) against the REFERENCE's outputs -- not against the
    repo's own full-save run -- in bf16 under the 1.5x rule; and the full-save run of the same case beside it gives the same numbers
    to bf16 round-off (the rebuilt LayerNorm / GELU outputs are the same kernels on the same inputs)."""
    monkeypatch.setenv("VITK_RECOMPUTE", "1")
    e, g, e16, g16, worst = _wide_errors("vit_h14_d4_b8", torch.bfloat16)
    print(f"vit_h14_d4_b8 bf16, recompute forced: logits {e:.2e} (reference-bf16 {e16:.2e}) grad samples {g:.2e} (reference-bf16 {g16:.2e}) worst tensor {worst:.2e}")
    assert e <= 1.5 * e16 + 1e-3 and g <= 1.5 * g16 + 1e-3, (e, e16, g, g16)
    assert worst <= 0.15, worst
    monkeypatch.setenv("VITK_RECOMPUTE", "0")
    e0, g0, _, _, _ = _wide_errors("vit_h14_d4_b8", torch.bfloat16)
    assert abs(e - e0) <= 1e-3 and abs(g - g0) <= 1e-3, (e, e0, g, g0)


VITB_SMALL = dict(image_size=224, patch_size=16, num_classes=1000, dim=768, depth=2, heads=12, mlp_dim=3072)


@pytest.mark.parametrize("kind", ["vit", "simple_vit"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_vit_b16_width_depth2_vs_oracle(kind, dtype):
    """BASELINE config 2's layer shapes (N=197, D=768, h=12, F=3072: the MFMA fast path in bf16)
    at depth 2 / batch 4 so the CPU oracle finishes in seconds."""
    cfg = dict(VITB_SMALL)
    params = make_params(kind, cfg, 7)
    img = make_images(cfg, 4, 1007)
    ref_out, ref_g = O.run_fwd_bwd(kind, cfg, params, img, torch.float32)
    out, grads = run_mine(kind, cfg, params, img, dtype, loss_scale=4096.0 if dtype == torch.float16 else 1.0)
    keys = list(params)
    cat = lambda d: torch.cat([d[k].detach().float().flatten().cpu() for k in keys])
    e, g = rel(out, ref_out), rel(cat(grads), cat(ref_g))
    if dtype == torch.float32:
        assert e <= 1e-3 and g <= 1e-3, (e, g)
        worst = max(rel(grads[k], ref_g[k]) for k in keys)
        assert worst <= 1e-3, worst
    elif dtype == torch.float16:                      # 11-bit significand: an absolute gate (see test_fp16_mode_vs_golden)
        print(f"{kind} fp16 depth2: logits {e:.2e} grads {g:.2e}")
        assert e <= 3e-3 and g <= 3e-3, (e, g)
    else:
        bf_out, bf_g = O.run_fwd_bwd(kind, cfg, params, img, torch.bfloat16)
        e_ref, g_ref = rel(bf_out, ref_out), rel(cat(bf_g), cat(ref_g))
        print(f"{kind} bf16 depth2: logits {e:.2e} (reference-bf16 {e_ref:.2e}); grads {g:.2e} (reference-bf16 {g_ref:.2e})")
        assert e <= 1.5 * e_ref + 1e-3, (e, e_ref)
        assert g <= 1.5 * g_ref + 1e-3, (g, g_ref)
        # no gradient tensor may be grossly off (catches a wrong-but-small tensor hidden in the concatenation)
        for k in keys:
            assert rel(grads[k], ref_g[k]) <= 0.12, (k, rel(grads[k], ref_g[k]))


def test_full_config2_size_properties():
    """ViT-B/16, depth 12, bf16 at batch 64: size-independent properties instead of an oracle run.
    (1) run-to-run bitwise determinism of logits and gradients; (2) permuting the batch permutes
    the logits bit-exactly and leaves every gradient unchanged up to f32 summation order."""
    cfg = dict(VITB_SMALL, depth=12)
    torch.manual_seed(0)
    m = ViT(**cfg).to(DEV, dtype=torch.bfloat16)
    img = torch.randn(64, 3, 224, 224, device=DEV).to(torch.bfloat16)

    def step(x):
        m.zero_grad(set_to_none=True)
        out = m(x)
        O.loss_fn(out).backward()
        return out.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters()}

    o1, g1 = step(img)
    o2, g2 = step(img)
    assert torch.isfinite(o1.float()).all()
    assert torch.equal(o1, o2)
    for k in g1:
        assert torch.isfinite(g1[k].float()).all(), k
        assert torch.equal(g1[k], g2[k]), k
    perm = torch.randperm(64, device=DEV)
    o3, g3 = step(img[perm])
    assert torch.equal(o3, o1[perm])
    for k in g1:
        assert rel(g3[k], g1[k]) <= 2e-2, (k, rel(g3[k], g1[k]))


def test_reference_unit_test_verbatim():
    """tests/test_vit.py:4-20 of the reference, on the GPU: train mode, dropout 0.1, shape check."""
    v = ViT(image_size=256, patch_size=32, num_classes=1000, dim=1024, depth=6, heads=16, mlp_dim=2048,
            dropout=0.1, emb_dropout=0.1).to(DEV)
    img = torch.randn(1, 3, 256, 256, device=DEV)
    preds = v(img)
    assert preds.shape == (1, 1000), 'correct logits outputted'
    preds.float().square().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in v.parameters())


def test_hooks_and_submodules_match_fused_path():
    """Recorder-style forward hook on `attend` (recorder.py:26-29) forces the materialising path;
    its result and gradients must agree with the fused path, and the hook must see (b,h,n,n)."""
    case = CASES["vit_cls_tiny"]
    params = make_params(case["kind"], case["cfg"], case["seed"])
    img = make_images(case["cfg"], case["batch"], case["seed"] + 1000)
    out_f, g_f = run_mine("vit", case["cfg"], params, img, torch.float32)
    m = build("vit", case["cfg"], params, torch.float32)
    seen = []
    hs = [layer[0].attend.register_forward_hook(lambda mod, i, o: seen.append(o.detach())) for layer in m.transformer.layers]
    out_h = m(img.to(DEV))
    O.loss_fn(out_h).backward()
    assert len(seen) == case["cfg"]["depth"]
    n = (32 // 8) ** 2 + 1
    assert tuple(seen[0].shape) == (case["batch"], case["cfg"]["heads"], n, n)
    assert torch.allclose(seen[0].sum(-1), torch.ones_like(seen[0].sum(-1)), atol=1e-5)
    assert rel(out_h, out_f) < 1e-5
    for k, p in m.named_parameters():
        if p.numel():
            assert rel(p.grad, g_f[k]) < 1e-4, k
    for h in hs:
        h.remove()
    # sub-modules are callable on their own (mae.py:74, simmim.py:70, mpp.py:169 do this)
    tokens = m.to_patch_embedding(img.to(DEV))
    assert tuple(tokens.shape) == (case["batch"], n - 1, case["cfg"]["dim"])
    y = m.transformer(tokens)
    assert tuple(y.shape) == tuple(tokens.shape)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("kind,case_name", [("vit", "vit_cls_tiny"), ("simple", "cfg1_simple_vit_tiny")])
def test_16bit_unfused_transformer_keeps_embedding_in_the_graph(kind, case_name, dtype):
    """A 16-bit model whose embedding runs fused (f32 residual stream out) while the transformer runs op by op (forward hook on
    `attend`): the dtype hand-over between the two is an autograd node, so cls / pos / patch-embedding parameters still get
    gradients, and they agree with the fully fused run."""
    case = CASES[case_name]
    params = make_params(case["kind"], case["cfg"], case["seed"])
    img = make_images(case["cfg"], case["batch"], case["seed"] + 1000)
    scale = 1024.0 if dtype == torch.float16 else 1.0
    out_f, g_f = run_mine(kind, case["cfg"], params, img, dtype, loss_scale=scale)
    m = build(kind, case["cfg"], params, dtype)
    hs = [layer[0].attend.register_forward_hook(lambda mod, i, o: None) for layer in m.transformer.layers]
    out_h = m(img.to(DEV, dtype=dtype))
    (O.loss_fn(out_h) * scale).backward()
    for h in hs:
        h.remove()
    assert rel(out_h, out_f) < 2e-2
    for k, p in m.named_parameters():
        if p.numel():
            assert p.grad is not None, f"{k} fell out of the graph"
            assert rel(p.grad.float() / scale, g_f[k]) < 0.1, k
    emb = [k for k, _ in m.named_parameters() if k.startswith(("to_patch_embedding", "cls_token", "pos_embedding"))]
    assert emb and all(g_f[k].abs().sum().item() > 0 for k in emb if g_f[k].numel())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_activation_recompute_matches_full_save(dtype, monkeypatch):
    """engine._recompute_policy: dropping the GELU output and the LayerNorm outputs and rebuilding them in backward (what lets
    ViT-H/14 at batch 256 fit) changes nothing in f32 and stays within 16-bit rounding otherwise (the rebuilt GELU is the
    elementwise kernel's erf form, the forward's the GEMM epilogue's polynomial: |diff| <= 2.2e-5 before rounding)."""
    cfg = dict(VITB_SMALL, depth=2)
    params = make_params("vit", cfg, 17)
    img = make_images(cfg, 6, 1017)
    monkeypatch.setenv("VITK_RECOMPUTE", "0")
    out0, g0 = run_mine("vit", cfg, params, img, dtype)
    monkeypatch.setenv("VITK_RECOMPUTE", "1")
    out1, g1 = run_mine("vit", cfg, params, img, dtype)
    assert torch.equal(out0, out1)
    # 16-bit: the full-save run stores the gelu' FACTOR (round 4, ops.gelu_dg_ok: one more 16-bit rounding inside dFF1), the recompute run the
    # pre-activation: 6e-3 between them; with VITK_GELU_DG=0 the two runs differ only by the rebuilt GELU: 3e-3 as before
    tol = 1e-6 if dtype == torch.float32 else 6e-3
    for k in g0:
        if g0[k].numel():
            assert rel(g1[k], g0[k]) <= tol, (k, rel(g1[k], g0[k]))
    if dtype != torch.float32:
        monkeypatch.setenv("VITK_GELU_DG", "0")
        monkeypatch.setenv("VITK_RECOMPUTE", "0")
        out2, g2 = run_mine("vit", cfg, params, img, dtype)
        assert torch.equal(out2, out1)
        for k in g2:
            if g2[k].numel():
                assert rel(g1[k], g2[k]) <= 3e-3, (k, rel(g1[k], g2[k]))


def test_cpu_input_fails_loudly():
    m = ViT(**CASES["vit_cls_tiny"]["cfg"])
    with pytest.raises(RuntimeError, match="HIP"):
        m(torch.randn(1, 3, 32, 32))


def test_batch_of_one_and_odd_batch():
    """Edge extents: a single image, and a batch that leaves ragged tails in every tile (M = 3 * 197)."""
    cfg = dict(VITB_SMALL, depth=1)
    params = make_params("vit", cfg, 11)
    for b in (1, 3):
        img = make_images(cfg, b, 2000 + b)
        ref_out, ref_g = O.run_fwd_bwd("vit", cfg, params, img, torch.float32)
        out, grads = run_mine("vit", cfg, params, img, torch.float32)
        assert rel(out, ref_out) <= 1e-3
        assert max(rel(grads[k], ref_g[k]) for k in params) <= 1e-3


@pytest.mark.parametrize("kind", ["vit", "simple_vit"])
@pytest.mark.parametrize("channels,patch", [(1, 4), (4, 8), (2, (4, 6))])
def test_channel_counts_and_small_patch_dims(kind, channels, patch):
    """`channels` other than 3 (vit.py:86, :94): patch_dim = 16 (below one MFMA K-step: zero-padded to 32), 256 and 48."""
    cfg = dict(image_size=(24, 24), patch_size=patch, num_classes=6, dim=64, depth=1, heads=2, dim_head=32, mlp_dim=96, channels=channels)
    params = make_params(kind, cfg, 21)
    img = make_images(cfg, 3, 2100)
    ref_out, ref_g = O.run_fwd_bwd(kind, cfg, params, img, torch.float32)
    for dtype, tol in ((torch.float32, 1e-3), (torch.bfloat16, 5e-2)):
        out, grads = run_mine(kind, cfg, params, img, dtype)
        assert rel(out, ref_out) <= tol, (dtype, rel(out, ref_out))
        keys = [k for k in params if params[k].numel()]
        cat = lambda d: torch.cat([d[k].detach().float().flatten().cpu() for k in keys])
        assert rel(cat(grads), cat(ref_g)) <= (tol if dtype == torch.float32 else 8e-2), dtype


def test_vit_l16_width_bf16_runs_fast_path():
    """BASELINE config 3's layer shapes (D=1024, h=16, F=4096) at depth 2 / batch 8, bf16, vs the f32 oracle."""
    cfg = dict(image_size=224, patch_size=16, num_classes=1000, dim=1024, depth=2, heads=16, mlp_dim=4096)
    params = make_params("vit", cfg, 13)
    img = make_images(cfg, 8, 1013)
    ref_out, ref_g = O.run_fwd_bwd("vit", cfg, params, img, torch.float32)
    bf_out, bf_g = O.run_fwd_bwd("vit", cfg, params, img, torch.bfloat16)
    out, grads = run_mine("vit", cfg, params, img, torch.bfloat16)
    keys = list(params)
    cat = lambda d: torch.cat([d[k].detach().float().flatten().cpu() for k in keys])
    e, e_ref = rel(out, ref_out), rel(bf_out, ref_out)
    g, g_ref = rel(cat(grads), cat(ref_g)), rel(cat(bf_g), cat(ref_g))
    print(f"vit-L width bf16: logits {e:.2e} (reference-bf16 {e_ref:.2e}); grads {g:.2e} (reference-bf16 {g_ref:.2e})")
    assert e <= 1.5 * e_ref + 1e-3 and g <= 1.5 * g_ref + 1e-3


def test_dropout_training_path_statistics():
    """dropout > 0 in train mode takes the op-by-op path; eval mode must equal the fused path exactly."""
    case = CASES["vit_cls_tiny"]
    cfg = dict(case["cfg"], dropout=0.2, emb_dropout=0.1)
    params = make_params("vit", case["cfg"], case["seed"])
    img = make_images(cfg, case["batch"], 77).to(DEV)
    m = ViT(**cfg)
    m.load_state_dict(params)
    m = m.to(DEV)
    m.eval()
    out_eval = m(img)
    out_ref, _ = run_mine("vit", case["cfg"], params, img.cpu(), torch.float32)
    assert rel(out_eval, out_ref) < 1e-5
    m.train()
    torch.manual_seed(5)
    o1 = m(img); o2 = m(img)
    assert not torch.equal(o1, o2)                      # fresh masks every call
    assert rel(o1, out_eval) > 1e-3                     # dropout is active
    o1.float().square().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters() if p.numel())


def test_vit_h14_shapes_bf16_fast_paths():
    """BASELINE config 5's shapes in bf16 (dim_head 80, N = 577 > 480, patch_dim 588 % 32 != 0): chunked attention
    kernels with the head dimension padded to 96 inside the MFMA K-steps, K-padded patch GEMM.  depth 1, batch 2."""
    cfg = dict(image_size=336, patch_size=14, num_classes=1000, dim=1280, depth=1, heads=16, dim_head=80, mlp_dim=5120)
    params = make_params("vit", cfg, 17)
    img = make_images(cfg, 2, 1017)
    ref_out, ref_g = O.run_fwd_bwd("vit", cfg, params, img, torch.float32)
    bf_out, bf_g = O.run_fwd_bwd("vit", cfg, params, img, torch.bfloat16)
    out, grads = run_mine("vit", cfg, params, img, torch.bfloat16)
    keys = list(params)
    cat = lambda d: torch.cat([d[k].detach().float().flatten().cpu() for k in keys])
    e, e_ref = rel(out, ref_out), rel(bf_out, ref_out)
    g, g_ref = rel(cat(grads), cat(ref_g)), rel(cat(bf_g), cat(ref_g))
    print(f"vit-H/14 shapes bf16: logits {e:.2e} (reference-bf16 {e_ref:.2e}); grads {g:.2e} (reference-bf16 {g_ref:.2e})")
    assert e <= 1.5 * e_ref + 1e-3 and g <= 1.5 * g_ref + 1e-3
    for k in keys:
        assert rel(grads[k], ref_g[k]) <= 0.15, (k, rel(grads[k], ref_g[k]))


def test_long_sequence_dim_head_64_uses_chunked_attention():
    """N = 577 with dim_head 64 (beyond the whole-head-in-LDS kernels): f32 oracle vs bf16 run, depth 1."""
    cfg = dict(image_size=384, patch_size=16, num_classes=10, dim=128, depth=1, heads=2, mlp_dim=256)
    params = make_params("vit", cfg, 19)
    img = make_images(cfg, 2, 1019)
    ref_out, ref_g = O.run_fwd_bwd("vit", cfg, params, img, torch.float32)
    bf_out, bf_g = O.run_fwd_bwd("vit", cfg, params, img, torch.bfloat16)
    out, grads = run_mine("vit", cfg, params, img, torch.bfloat16)
    keys = list(params)
    cat = lambda d: torch.cat([d[k].detach().float().flatten().cpu() for k in keys])
    assert rel(out, ref_out) <= 1.5 * rel(bf_out, ref_out) + 1e-3
    assert rel(cat(grads), cat(ref_g)) <= 1.5 * rel(cat(bf_g), cat(ref_g)) + 1e-3


@pytest.mark.parametrize("dh", [32, 48, 96])
def test_other_head_widths_run_on_the_flash_kernels(dh):
    """vit.py:86 leaves dim_head free: 32 / 48 / 96 (beside 64 and 80) take the chunked flash kernels in 16 bit (round 3; they
    used to fall to the kernels that materialise the attention matrix).  Oracle f32 vs bf16 run, depth 2, N = 65."""
    from vit_pytorch_amd import ops
    assert ops.attn_varlen_ok(torch.bfloat16, dh) and not ops.attn_fast_ok(torch.bfloat16, 65, dh)
    cfg = dict(image_size=64, patch_size=8, num_classes=10, dim=128, depth=2, heads=2, dim_head=dh, mlp_dim=256)
    params = make_params("vit", cfg, 23 + dh)
    img = make_images(cfg, 3, 1023 + dh)
    ref_out, ref_g = O.run_fwd_bwd("vit", cfg, params, img, torch.float32)
    bf_out, bf_g = O.run_fwd_bwd("vit", cfg, params, img, torch.bfloat16)
    out, grads = run_mine("vit", cfg, params, img, torch.bfloat16)
    keys = list(params)
    cat = lambda d: torch.cat([d[k].detach().float().flatten().cpu() for k in keys])
    assert rel(out, ref_out) <= 1.5 * rel(bf_out, ref_out) + 1e-3
    assert rel(cat(grads), cat(ref_g)) <= 1.5 * rel(cat(bf_g), cat(ref_g)) + 1e-3
    for k in keys:
        assert rel(grads[k], ref_g[k]) <= 0.15, (k, rel(grads[k], ref_g[k]))


def test_fused_dropout_layer_matches_masked_reference():
    """Active dropout (vit.py:22,24,48,60 at p = 0.1, training mode) runs inside the fused engine; its keep decisions are
    a counter hash, so a float64 reference that applies the very same masks (vitk_dropout_keep) must agree."""
    from vit_pytorch_amd import engine as E, kernels as K
    from vit_pytorch_amd.vit import Transformer
    B, N, D, H, d, Fh, p = 8, 197, 768, 12, 64, 3072, 0.1
    torch.manual_seed(5)
    blk = Transformer(D, 1, H, d, Fh, dropout=p).to(DEV, dtype=torch.bfloat16).train()
    for q in blk.parameters():                                   # non-trivial affine / bias values
        if q.ndim == 1:
            q.data.add_(0.1 * torch.randn_like(q))
    x = torch.randn(B, N, D, device=DEV).to(torch.bfloat16)
    assert blk._fusable(x) and blk._dropout_p() == p
    blk._drop_calls = 3
    seed = (int(torch.initial_seed()) + 0x9E3779B1 * 3 + 0x85EBCA6B * blk._drop_salt) & 0xffffffff
    y = blk(x)
    assert blk._drop_calls == 4                                 # the fused path drew its seeds
    O.loss_fn(y).backward()

    def keep(rows, cols, k):
        m = torch.empty(rows, cols, dtype=torch.uint8, device=DEV)
        K.dropout_keep(m, rows, cols, p, E._hash32(seed + k))
        return m.double() / (1 - p)

    attn, ff = blk.layers[0]
    P = {k_: v.detach().double().requires_grad_(True) for k_, v in blk.named_parameters()}
    xd = x.double()
    ln = torch.nn.functional.layer_norm
    a1 = ln(xd, (D,), P["layers.0.0.norm.weight"], P["layers.0.0.norm.bias"], 1e-5)
    qkv = a1 @ P["layers.0.0.to_qkv.weight"].t()
    q, k, v = (qkv[..., i * H * d:(i + 1) * H * d].reshape(B, N, H, d).permute(0, 2, 1, 3) for i in range(3))
    pm = torch.softmax(q @ k.transpose(-1, -2) * d ** -0.5, -1) * keep(B * H * N, N, 0).view(B, H, N, N)
    o = (pm @ v).permute(0, 2, 1, 3).reshape(B, N, H * d)
    x2 = xd + (o @ P["layers.0.0.to_out.0.weight"].t() + P["layers.0.0.to_out.0.bias"]) * keep(B * N, D, 1).view(B, N, D)
    a2 = ln(x2, (D,), P["layers.0.1.net.0.weight"], P["layers.0.1.net.0.bias"], 1e-5)
    act = torch.nn.functional.gelu(a2 @ P["layers.0.1.net.1.weight"].t() + P["layers.0.1.net.1.bias"]) * keep(B * N, Fh, 2).view(B, N, Fh)
    x3 = x2 + (act @ P["layers.0.1.net.4.weight"].t() + P["layers.0.1.net.4.bias"]) * keep(B * N, D, 3).view(B, N, D)
    yref = ln(x3, (D,), P["norm.weight"], P["norm.bias"], 1e-5)
    O.loss_fn(yref).backward()
    e = rel(y, yref)
    keys = list(P)
    g = rel(torch.cat([blk.get_parameter(k_).grad.float().flatten() for k_ in keys]), torch.cat([P[k_].grad.flatten() for k_ in keys]))
    worst = max(rel(blk.get_parameter(k_).grad, P[k_].grad) for k_ in keys)
    print(f"fused dropout layer: out {e:.2e}, grads {g:.2e}, worst tensor {worst:.2e}")
    assert e < 1e-2 and g < 2e-2 and worst < 6e-2
    blk.eval()
    assert blk._dropout_p() == 0.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("kind", ["vit", "simple_vit"])
def test_a_batch_of_zero_images(kind, dtype):
    """vit.py:118-138 is shape-agnostic in the batch: an empty batch gives (0, classes) logits and zero gradients (missing 5)."""
    case = CASES["vit_cls_tiny" if kind == "vit" else "cfg1_simple_vit_tiny"]
    m = (ViT if kind == "vit" else SimpleViT)(**case["cfg"]).to("cuda").to(dtype)
    img = torch.zeros((0, 3, 32, 32), device="cuda", dtype=dtype, requires_grad=True)
    out = m(img)
    assert tuple(out.shape) == (0, case["cfg"]["num_classes"]) and out.dtype == dtype
    out.float().sum().backward()
    assert tuple(img.grad.shape) == (0, 3, 32, 32)
    for k, p in m.named_parameters():
        assert p.grad is not None and float(p.grad.float().abs().sum()) == 0.0, k
    if dtype == torch.float32:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            assert tuple(m(img).shape) == (0, case["cfg"]["num_classes"])
