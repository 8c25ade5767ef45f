"""CPU: the host logic of the fused Transformer stage (engine.TransformerFn) with the kernels replaced by test doubles
(tests/_kernel_doubles.py -- the oracle's per-op restatements in the kernels' calling conventions).  What is checked here is
the PLUMBING: operand routing, saved activations, the 16-bit gradient stream, the activation-recompute policy and the fp8
state machine (recording passes, delayed scales, which GEMM runs in which format).  The kernels themselves are checked on the GPU."""
import os
from collections import OrderedDict

import pytest
import torch

from oracle import vit_oracle as O
from oracle.params import make_params_for
from vit_pytorch_amd import _lib as L
from vit_pytorch_amd import engine as E
from vit_pytorch_amd import kernels as K
from vit_pytorch_amd import ops
from vit_pytorch_amd.fp8 import SLOTS_PER_LAYER, enable_fp8, enable_fp8_forward
from vit_pytorch_amd.vit import Transformer

import _kernel_doubles as KD

DIM, DEPTH, HEADS, DH, MLP = 256, 2, 4, 64, 512
B, N = 8, 128                     # M = 1024: the smallest extent the 256-row GEMM kernels (and with them the fp8 path) serve


def rel(a, b):
    a = a.detach().double().flatten(); b = b.detach().double().flatten()
    return ((a - b).norm() / b.norm()).item()


def build(dtype):
    m = Transformer(DIM, DEPTH, HEADS, DH, MLP)
    shapes = OrderedDict((k, tuple(v.shape)) for k, v in m.state_dict().items())
    params = make_params_for(shapes, 11)
    m.load_state_dict(params)
    return m.to(dtype), params


def reference(params, x):
    """The oracle's Transformer.forward (vit.py:78-83) in float32 under torch autograd."""
    p = {"transformer." + k: v.clone().requires_grad_(True) for k, v in params.items()}
    xr = x.clone().requires_grad_(True)
    y = O.transformer_fwd(xr, p, DEPTH, HEADS, DH, simple=False)
    O.loss_fn(y).backward()
    return y.detach(), xr.grad, {k[len("transformer."):]: v.grad for k, v in p.items()}


def run(m, x):
    m.zero_grad(set_to_none=True)
    xi = x.clone().requires_grad_(True)
    y = m(xi)
    O.loss_fn(y).backward()
    return y.detach().float(), xi.grad.float(), {k: p.grad.float() for k, p in m.named_parameters()}


@pytest.fixture()
def x():
    g = torch.Generator().manual_seed(5)
    return torch.randn(B, N, DIM, generator=g)


def worst_grad(g, gref):
    return max(rel(g[k], gref[k]) for k in gref)


def test_16bit_stage_against_the_oracle(x):
    m, params = build(torch.bfloat16)
    y_ref, dx_ref, g_ref = reference(params, x)
    with KD.installed() as calls:
        y, dx, g = run(m, x)
    assert rel(y, y_ref) < 2e-2 and rel(dx, dx_ref) < 4e-2 and worst_grad(g, g_ref) < 6e-2, (rel(y, y_ref), rel(dx, dx_ref), worst_grad(g, g_ref))
    names = [c[0] for c in calls]
    assert names.count("gemm_tn_bf16") == 4 * DEPTH                     # dW of QKV, out, FF1, FF2
    assert "gemm_nt_fp8_v2" not in names and "quantize_fp8_delayed" not in names


def test_16bit_forward_stream_option(x, monkeypatch):
    """VITK_FWD_STREAM=16 (round 4: the default for bfloat16) vs =f32: the residual stream of the forward in the parameter dtype -- both residual GEMMs of a layer take the RESID16
    epilogue (16-bit residual in, 16-bit sum out), LayerNorm reads the 16-bit stream, no float32 (M, D) tensor is produced."""
    m, params = build(torch.bfloat16)
    y_ref, dx_ref, g_ref = reference(params, x)
    with KD.installed() as calls:
        monkeypatch.setenv("VITK_FWD_STREAM", "f32")
        y32, dx32, g32 = run(m, x)
        n32 = [c[1][3] for c in calls if c[0] == "gemm_nt_bf16"]
        assert n32.count(L.EPI_RESID) == 2 * DEPTH and n32.count(L.EPI_RESID16) == 0
        del calls[:]
        monkeypatch.setenv("VITK_FWD_STREAM", "16")
        y, dx, g = run(m, x)
        n16 = [c[1][3] for c in calls if c[0] == "gemm_nt_bf16"]
        assert n16.count(L.EPI_RESID16) == 2 * DEPTH and n16.count(L.EPI_RESID) == 0
    assert not torch.equal(y, y32)
    assert rel(y, y_ref) < 3e-2 and rel(dx, dx_ref) < 6e-2 and worst_grad(g, g_ref) < 8e-2, (rel(y, y_ref), rel(dx, dx_ref), worst_grad(g, g_ref))
    print(f"16-bit forward stream vs f32 oracle: out {rel(y, y_ref):.2e} (f32 stream {rel(y32, y_ref):.2e}), worst grad {worst_grad(g, g_ref):.2e} ({worst_grad(g32, g_ref):.2e})")


@pytest.mark.parametrize("dtype,env,codes", [(torch.bfloat16, None, True), (torch.bfloat16, "16", False), (torch.float16, None, False), (torch.bfloat16, "0", None)])
def test_gelu_factor_storage_between_ff1_and_dff1(x, dtype, env, codes, monkeypatch):
    """What FF1 leaves for dFF1 (ops.linear_fwd(save_dg=True) / ops.linear_dx(gelu_dg=...)): 8-bit fixed-point codes of gelu'(pre) in a
    bfloat16 model (EPI_BIAS_GELU_DG8 -> vitk_gemm_nt_bf16_mul_aux8_colsum), the 16-bit factor in a float16 model or under VITK_GELU_DG=16
    (EPI_BIAS_GELU_DG -> _mul_aux_colsum), the pre-activation under VITK_GELU_DG=0 (EPI_BIAS_GELU -> _gelu_bwd_colsum).  The three agree
    with the oracle to the same gates."""
    if env is None:
        monkeypatch.delenv("VITK_GELU_DG", raising=False)
    else:
        monkeypatch.setenv("VITK_GELU_DG", env)
    m, params = build(dtype)
    y_ref, dx_ref, g_ref = reference(params, x)
    with KD.installed() as calls:
        y, dx, g = run(m, x)
    epis = [c[1][3] for c in calls if c[0] == "gemm_nt_bf16"]
    names = [c[0] for c in calls]
    if codes is None:
        assert epis.count(L.EPI_BIAS_GELU) == DEPTH and L.EPI_BIAS_GELU_DG not in epis and L.EPI_BIAS_GELU_DG8 not in epis
        assert names.count("gemm_nt_bf16_gelu_bwd_colsum") == DEPTH
    elif codes:
        assert epis.count(L.EPI_BIAS_GELU_DG8) == DEPTH and L.EPI_BIAS_GELU_DG not in epis and L.EPI_BIAS_GELU not in epis
        assert names.count("gemm_nt_bf16_mul_aux8_colsum") == DEPTH and "gemm_nt_bf16_mul_aux_colsum" not in names
    else:
        assert epis.count(L.EPI_BIAS_GELU_DG) == DEPTH and L.EPI_BIAS_GELU_DG8 not in epis
        assert names.count("gemm_nt_bf16_mul_aux_colsum") == DEPTH and "gemm_nt_bf16_mul_aux8_colsum" not in names
    # (float16: this loss leaves gradients of 1e-7, below the format's normal range -- the gates are the bfloat16 ones)
    assert rel(y, y_ref) < 2e-2 and rel(dx, dx_ref) < 4e-2 and worst_grad(g, g_ref) < 9e-2, (rel(y, y_ref), rel(dx, dx_ref), worst_grad(g, g_ref))


def fp8_calls(calls):
    return [c[1] for c in calls if c[0] == "gemm_nt_fp8_v2"]


@pytest.mark.parametrize("mode", ["fwd+dx+dw", "fwd+dx+dw/requantise", "fwd+dx", "fwd"])
def test_fp8_state_machine_and_formats(x, mode, monkeypatch):
    monkeypatch.setenv("VITK_FP8_K128", "0")
    monkeypatch.setenv("VITK_FWD_STREAM", "f32")      # the fp8 path keeps the float32 stream: its recording step equals the 16-bit run under that stream
    monkeypatch.setenv("VITK_GELU_DG", "0")           # ... and saves the pre-activation (the 16-bit default stores the gelu' factor instead)
    lean = mode == "fwd+dx+dw"             # the forward's e4m3 copies are kept for the weight-gradient GEMMs (the default)
    monkeypatch.setattr(E, "FP8_LEAN", bool(lean))
    backward, wgrad = mode != "fwd", mode.startswith("fwd+dx+dw")
    m16, params = build(torch.bfloat16)
    y_ref, dx_ref, g_ref = reference(params, x)
    m8, _ = build(torch.bfloat16)
    if backward:
        enable_fp8(m8, wgrad=wgrad)
    else:
        enable_fp8_forward(m8)
    st = m8._fp8
    assert st.backward is backward and st.wgrad is wgrad and not st.ready and not st.bwd_ready and not st.k128
    M, I = B * N, HEADS * DH
    with KD.installed() as calls:
        y16, dx16, g16 = run(m16, x)
        del calls[:]
        # step 1: every GEMM in 16 bit, amax of the four activation tensors (forward) and the four gradient tensors (backward) recorded
        y1, dx1, g1 = run(m8, x)
        assert torch.equal(y1, y16) and torch.equal(dx1, dx16) and all(torch.equal(g1[k], g16[k]) for k in g16)
        assert not fp8_calls(calls)
        rec = [c[1] for c in calls if c[0] == "quantize_fp8_delayed"]
        assert all(not r[1] and r[2] for r in rec)                       # record only, nothing written
        assert len(rec) == DEPTH * (1 + (4 if backward else 0))          # attention output + four gradients per layer
        assert st.ready and not st.bwd_ready
        sc = st.scales.view(DEPTH, SLOTS_PER_LAYER, 2)
        assert (sc[:, :4, 0] > 0).all() and float(sc[:, 4:].abs().sum()) == 0.0          # activation scales decided, gradient scales not yet
        assert torch.allclose(sc[:, :4, 0] * sc[:, :4, 1], torch.ones(DEPTH, 4), rtol=1e-5)
        am = st.amax.view(DEPTH, SLOTS_PER_LAYER, 64)
        assert int(am[:, :4].abs().sum()) == 0                           # forward records folded and reset
        assert (int(am[:, 4:].abs().sum()) > 0) is backward              # backward records wait for the next fold
        del calls[:]
        # step 2: forward on e4m3 operands; the fold after it decides the gradient scales, so this backward runs on e5m2 gradients
        y2, dx2, g2 = run(m8, x)
        f = fp8_calls(calls)
        fwd = [c for c in f if c[4] == K.A_E4M3]
        bwd = [c for c in f if c[4] == K.A_E5M2]
        assert len(fwd) == 4 * DEPTH and len(bwd) == (4 * DEPTH if backward else 0)
        per_layer_fwd = {(M, 3 * I, DIM, L.EPI_NONE), (M, DIM, I, L.EPI_RESID), (M, MLP, DIM, L.EPI_BIAS_GELU), (M, DIM, MLP, L.EPI_RESID)}
        assert {c[:4] for c in fwd} == per_layer_fwd
        if backward:
            per_layer_bwd = {(M, MLP, DIM, L.EPI_GELU_BWD), (M, DIM, MLP, L.EPI_NONE), (M, I, DIM, L.EPI_NONE), (M, DIM, 3 * I, L.EPI_NONE)}
            assert {c[:4] for c in bwd} == per_layer_bwd
            assert st.bwd_ready and (st.scales.view(DEPTH, SLOTS_PER_LAYER, 2)[:, 4:, 0] > 0).all()
            # the weight-gradient GEMMs: on the e5m2 copies the dX GEMMs already made + e4m3 copies of the saved activations, or 16 bit
            names = [c[0] for c in calls]
            assert names.count("gemm_tn_fp8") == (4 * DEPTH if wgrad else 0) and names.count("gemm_tn_bf16") == (0 if wgrad else 4 * DEPTH)
            if wgrad:
                assert {c[1][:3] for c in calls if c[0] == "gemm_tn_fp8"} == {(M, DIM, MLP), (M, MLP, DIM), (M, DIM, I), (M, 3 * I, DIM)}
                qs = [c[1] for c in calls if c[0] == "quantize_fp8_delayed"]
                # per layer: o (fwd, e4m3, recorded), 4 gradients (e5m2, recorded); the e4m3 activation operands are the copies the
                # forward kept (lean) or the 4 saved 16-bit activations re-quantised (e4m3, not recorded)
                requant = sum(1 for q in qs if q[3] == K.FMT_E4M3 and not q[2])
                # (the e5m2 copy of dpre comes out of the GELU' epilogue: 3 gradient passes a layer, not 4)
                assert (len(qs), requant) == ((4 * DEPTH, 0) if lean else (8 * DEPTH, 4 * DEPTH))
            assert not [c for c in calls if c[0] in ("gemm_nt_bf16", "gemm_nt_bf16_gelu_bwd_colsum")]
        assert not any(c[5] for c in f)                                  # VITK_FP8_K128=0: the K = 32 forms
        # numerics of the plumbing: fp8-sized distance from the 16-bit run and from the f32 oracle
        e_y, e_dx, e_g = rel(y2, y16), rel(dx2, dx16), worst_grad(g2, g16)
        print(f"fp8 ({mode}) vs 16-bit: out {e_y:.2e} dx {e_dx:.2e} worst grad {e_g:.2e}; vs f32 oracle: out {rel(y2, y_ref):.2e}")
        assert 1e-4 < e_y < 6e-2 and e_dx < 1.2e-1 and e_g < 1.5e-1 and rel(y2, y_ref) < 6e-2
        # unchanged weights are not re-quantised; an in-place update is
        n_w, n_wt = len(st._w), len(st._wt)
        assert n_w == 4 * DEPTH and n_wt == (4 * DEPTH if backward else 0)
        key0 = st._w[id(m8.layers[0][1].net[1].weight)][0]
        run(m8, x)
        assert st._w[id(m8.layers[0][1].net[1].weight)][0] == key0 and len(st._w) == n_w
        with torch.no_grad():
            m8.layers[0][1].net[1].weight.mul_(1.0)
        run(m8, x)
        assert st._w[id(m8.layers[0][1].net[1].weight)][0] != key0


def test_fp8_k128_switch_and_recompute(x, monkeypatch):
    monkeypatch.delenv("VITK_FP8_K128", raising=False)             # the default: K = 128 wherever the reduction extent allows
    m8, _ = build(torch.bfloat16)
    enable_fp8(m8)
    assert m8._fp8.k128
    with KD.installed() as calls:
        run(m8, x); run(m8, x)
        del calls[:]
        y_a, dx_a, g_a = run(m8, x)
        f = fp8_calls(calls)
        assert len(f) == 8 * DEPTH and all(c[5] == (c[2] % 128 == 0) for c in f) and any(c[5] for c in f)
        tn = [c[1] for c in calls if c[0] == "gemm_tn_fp8"]
        assert len(tn) == 4 * DEPTH and all(c[3] for c in tn)             # every weight-gradient GEMM takes the K = 128 form (tokens are zero-padded)
        monkeypatch.setattr(E, "FP8_LEAN", False)                           # the recompute policy matters where 16-bit activations are saved
        # the activation-recompute policy (engine._recompute_policy, what lets ViT-H/14 batch 256 fit) composes with fp8: the backward
        # rebuilds the LayerNorm / GELU outputs it no longer finds saved (the delayed scales moved by one step in between, so the
        # two runs agree to quantisation noise, not bit for bit)
        monkeypatch.setenv("VITK_RECOMPUTE", "1")
        y_b, dx_b, g_b = run(m8, x)
        assert rel(y_b, y_a) < 2e-2 and rel(dx_b, dx_a) < 6e-2 and worst_grad(g_b, g_a) < 1e-1     # a routing error shows as O(1)
        # round 6: with the lean saving ON the same switch (or a stack that would not fit: _recompute_policy with frac = 0.70) makes the
        # lean saving give way to the recompute -- no e4m3 activation is kept, the weight-gradient GEMMs re-make their operands
        monkeypatch.setattr(E, "FP8_LEAN", True)
        kept = []
        orig = E._recompute_policy
        monkeypatch.setattr(E, "_recompute_policy", lambda b, d, frac=0.45: (kept.append(frac), orig(b, d, frac))[1])
        y_c, dx_c, g_c = run(m8, x)
        assert 0.70 in kept                                                  # the lean question was asked ...
        assert rel(y_c, y_a) < 2e-2 and rel(dx_c, dx_a) < 6e-2 and worst_grad(g_c, g_a) < 1e-1
        monkeypatch.setenv("VITK_RECOMPUTE", "0")                           # ... and with the switch off the lean saving stays
        y_d, dx_d, g_d = run(m8, x)
        assert rel(y_d, y_a) < 2e-2 and worst_grad(g_d, g_a) < 1e-1


def test_fp8_needs_16bit_operands_and_a_transformer(x):
    # a float32 model may be switched (its 16-bit operands exist under torch.autocast: functional.autocast_aware), but a float32
    # FORWARD with the switch on raises instead of silently ignoring it
    m, _ = build(torch.float32)
    enable_fp8(m)
    assert m._fp8 is not None
    with KD.installed(), pytest.raises(L.VitkError, match="autocast"):
        m(x.float())
    with pytest.raises(L.VitkError):
        enable_fp8(torch.nn.Linear(4, 4).to(torch.bfloat16))
    m16, _ = build(torch.bfloat16)
    enable_fp8(m16)
    assert m16._fp8 is not None
    enable_fp8(m16, enabled=False)
    assert m16._fp8 is None


def test_fp8_on_the_simple_vit_stack(x, monkeypatch):
    """simple_vit.Transformer (bias-free out-projection, simple_vit.py:48) through the same fp8 state machine: all twelve GEMMs of a
    layer on fp8 operands from the second step on, results fp8-close to its own 16-bit run."""
    from vit_pytorch_amd.simple_vit import Transformer as SimpleTransformer
    monkeypatch.delenv("VITK_FP8_K128", raising=False)

    def make():
        torch.manual_seed(3)
        m = SimpleTransformer(DIM, DEPTH, HEADS, DH, MLP)
        shapes = OrderedDict((k, tuple(v.shape)) for k, v in m.state_dict().items())
        m.load_state_dict(make_params_for(shapes, 13))
        return m.to(torch.bfloat16)

    with KD.installed() as calls:
        y16, dx16, g16 = run(make(), x)
        m8 = make()
        enable_fp8(m8)
        run(m8, x); run(m8, x)
        del calls[:]
        y8, dx8, g8 = run(m8, x)
        names = [c[0] for c in calls]
        assert names.count("gemm_nt_fp8_v2") == 8 * DEPTH and names.count("gemm_tn_fp8") == 4 * DEPTH
        assert "gemm_tn_bf16" not in names and "gemm_nt_bf16" not in names
    assert set(g8) == set(g16) and all(v is not None for v in g8.values())
    assert rel(y8, y16) < 6e-2 and rel(dx8, dx16) < 1.2e-1 and worst_grad(g8, g16) < 1.5e-1


def test_a_step_packs_the_weights_of_the_stack_in_one_table_call(x, monkeypatch):
    """ops.prepack_weights (round 6): the K-blocked copies of a step go out as ONE vitk_pack_w_nt_many table at the top of the stage --
    forward and transposed packs of the four Linear weights of every layer -- and a second step on unchanged weights packs nothing."""
    monkeypatch.setattr(ops, "_persistent_nt", lambda M, N, Kd: True)
    m, _ = build(torch.bfloat16)
    with KD.installed() as calls:
        run(m, x)
        tables = [c for c in calls if c[0] == "pack_w_nt_many"]
        assert [c[1] for c in tables] == [4 * DEPTH], tables
        del calls[:]
        run(m, x)
        assert not [c for c in calls if c[0] == "pack_w_nt_many"]
        with torch.no_grad():
            for p in m.parameters():
                p.add_(0.0)                                  # an optimizer step: version counters move
        del calls[:]
        run(m, x)
        assert [c[1] for c in calls if c[0] == "pack_w_nt_many"] == [4 * DEPTH]
