"""CPU: the seeded random configurations of tests/test_fuzz_gpu.py through the drop-in's HOST LOGIC with the kernels replaced by the
test doubles (tests/_kernel_doubles.py: the oracle's per-op restatements in the kernels' calling conventions), in float32 against the
oracle -- which tests/test_oracle_fuzz_vs_reference.py holds to the reference itself on these very draws.  What this covers without a
GPU: the dispatch between the fused stages and the op-by-op path (patch_dim 147 -> op by op), row maps / cls / positional handling at
rectangular patches and 1 / 4 channels, pooling, which gradient lands where, the image gradient, a model called twice, NaViT's
packing at ragged sizes and free dim_head.  Tolerance: round-off (logits 2e-5, gradients 2e-4 of the concatenated vector)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import _kernel_doubles as KD  # noqa: E402
import test_fuzz_gpu as F  # noqa: E402  (the draws)
from oracle import navit_oracle as NO  # noqa: E402
from oracle import vit_oracle as O  # noqa: E402
from oracle.params import make_images, make_navit_images, make_navit_params, make_params  # noqa: E402
from vit_pytorch_amd import SimpleViT, ViT  # noqa: E402
from vit_pytorch_amd.na_vit import NaViT  # noqa: E402


def rel(a, b):
    a = a.detach().double().flatten(); b = b.detach().double().flatten()
    n = b.norm().item()
    return (a - b).norm().item() / (n if n > 0 else 1.0)


@pytest.fixture(autouse=True)
def _plain_f32_routes(monkeypatch):
    # float32 on the MFMA kernels (three-term bf16 split, ops.f32_on_mfma) and the hi + lo flash attention are compositions of GPU kernels
    # without doubles; the plain float32 routes run the same host logic
    monkeypatch.setenv("VITK_F32_MFMA", "0")


@pytest.mark.parametrize("seed", range(F.N_DRAWS))
def test_fuzz_draw_host_logic_f32(seed):
    kind, cfg, batch = F.draw(seed)
    batch = min(batch, 6)                       # host logic does not depend on the batch; keep the CPU suite short
    params = make_params(kind, cfg, 50 + seed)
    img = make_images(cfg, batch, 1050 + seed)
    ref_out, ref_g = O.run_fwd_bwd(kind, cfg, params, img, torch.float32)
    m = (ViT if kind == "vit" else SimpleViT)(**cfg)
    m.load_state_dict(params, strict=True)
    x = img.clone().requires_grad_(seed % 3 == 0)          # a third of the draws also ask for the image gradient
    with KD.installed():
        out = m(x)
        O.loss_fn(out).backward()
    keys = [k for k in params if params[k].numel()]
    grads = {k: p.grad for k, p in m.named_parameters()}
    cat = lambda d: torch.cat([d[k].detach().float().flatten() for k in keys])
    assert rel(out, ref_out) <= 2e-5 and rel(cat(grads), cat(ref_g)) <= 2e-4, (kind, cfg, batch)
    if x.requires_grad:
        p64 = {k: v.double() for k, v in params.items()}
        x64 = img.double().requires_grad_(True)
        kw = dict(patch_size=cfg["patch_size"], depth=cfg["depth"], heads=cfg["heads"], dim_head=cfg["dim_head"])
        if kind == "vit":
            kw.update(pool=cfg["pool"], num_classes=cfg["num_classes"])
        O.loss_fn((O.vit_fwd if kind == "vit" else O.simple_vit_fwd)(x64, p64, **kw)).backward()
        assert x.grad is not None and rel(x.grad, x64.grad) <= 2e-4, (kind, cfg, rel(x.grad, x64.grad))


@pytest.mark.parametrize("seed", range(8))
def test_navit_fuzz_draw_host_logic_f32(seed):
    cfg, packs = F.draw_navit(seed)
    params = make_navit_params(cfg, 90 + seed)
    images = make_navit_images(cfg, packs, 1090 + seed)
    ref_out, ref_g = NO.run_fwd_bwd(cfg, params, images, torch.float32)
    m = NaViT(**cfg)
    m.load_state_dict(params, strict=True)
    m.eval()
    with KD.installed():
        out = m(images)
        O.loss_fn(out).backward()
    keys = [k for k in ref_g if ref_g[k].numel()]
    cat = lambda d: torch.cat([d[k].detach().float().flatten() for k in keys])
    assert rel(out, ref_out) <= 2e-5 and rel(cat({k: p.grad for k, p in m.named_parameters()}), cat(ref_g)) <= 2e-4, (cfg, packs)


def test_model_called_twice_and_accumulation_host_logic():
    kind, cfg, _ = F.draw(4)
    params = make_params(kind, cfg, 54)
    xa, xb = make_images(cfg, 3, 1), make_images(cfg, 2, 2)
    la = lambda o: o.float().square().mean()
    lb = lambda o: (o.float() - 1).square().mean()

    def fresh():
        m = ViT(**cfg); m.load_state_dict(params)
        return m

    g = lambda m: torch.cat([p.grad.flatten() for p in m.parameters() if p.numel()])
    with KD.installed():
        m = fresh(); la(m(xa)).backward(); ga = g(m)
        m = fresh(); lb(m(xb)).backward(); gb = g(m)
        m = fresh(); (la(m(xa)) + lb(m(xb))).backward(); both = g(m)
        m = fresh(); la(m(xa)).backward(); lb(m(xb)).backward(); acc = g(m)
    assert rel(both, ga + gb) <= 1e-5 and rel(acc, ga + gb) <= 1e-5


@pytest.mark.parametrize("pattern", ["head_only", "last_layer_and_head", "norms_and_biases", "embedding_only"])
def test_partially_frozen_models_host_logic(pattern):
    """Linear probing / partial fine-tuning: `requires_grad_(False)` on parts of the model.  The trainable parameters get the gradients
    the fully trainable model computes for them, the frozen ones get none, and nothing upstream of the lowest trainable parameter is
    required to exist."""
    cfg = dict(image_size=32, patch_size=8, num_classes=7, dim=32, depth=3, heads=2, dim_head=16, mlp_dim=64, pool="cls")
    params = make_params("vit", cfg, 77)
    img = make_images(cfg, 4, 1077)
    _, ref_g = O.run_fwd_bwd("vit", cfg, params, img, torch.float32)
    m = ViT(**cfg)
    m.load_state_dict(params, strict=True)
    train = {"head_only": lambda n: n.startswith("mlp_head"),
             "last_layer_and_head": lambda n: n.startswith(("mlp_head", "transformer.layers.2", "transformer.norm")),
             "norms_and_biases": lambda n: "norm" in n or n.endswith(".bias"),
             "embedding_only": lambda n: n.startswith(("to_patch_embedding", "cls_token", "pos_embedding"))}[pattern]
    for n, p in m.named_parameters():
        p.requires_grad_(bool(train(n)))
    with KD.installed():
        out = m(img)
        O.loss_fn(out).backward()
    for n, p in m.named_parameters():
        if train(n) and p.numel():
            assert p.grad is not None and rel(p.grad, ref_g[n]) <= 2e-4, (pattern, n)
        elif not train(n):
            assert p.grad is None, (pattern, n)


@pytest.mark.parametrize("reentrant", [False, True])
def test_torch_checkpoint_around_the_transformer_host_logic(reentrant):
    """torch.utils.checkpoint around the fused Transformer stage (a memory knob users of the reference reach for): same logits and
    gradients as the plain call, in both flavours (the reentrant one runs the first forward under no_grad -- the stage then keeps
    nothing -- and a second one inside backward)."""
    from torch.utils.checkpoint import checkpoint
    cfg = dict(image_size=32, patch_size=8, num_classes=7, dim=32, depth=2, heads=2, dim_head=16, mlp_dim=64, pool="mean")
    params = make_params("vit", cfg, 78)
    img = make_images(cfg, 3, 1078)
    _, ref_g = O.run_fwd_bwd("vit", cfg, params, img, torch.float32)

    class Ckpt(torch.nn.Module):
        def __init__(self, inner):
            super().__init__()
            self.inner = inner

        def forward(self, x):
            return checkpoint(self.inner, x, use_reentrant=reentrant)

    m = ViT(**cfg)
    m.load_state_dict(params, strict=True)
    with KD.installed():
        plain = m(img)
        m.transformer = Ckpt(m.transformer)
        out = m(img)
        O.loss_fn(out).backward()
    assert torch.equal(out, plain)
    for n, p in m.named_parameters():
        if p.numel():
            assert p.grad is not None and rel(p.grad, ref_g[n.replace("transformer.inner.", "transformer.")]) <= 2e-4, n


@pytest.mark.parametrize("kind", ["vit", "simple_vit"])
def test_swapped_heads_host_logic(kind):
    """`model.mlp_head = nn.Identity()` (features) and a fresh nn.Linear / nn.Sequential head (fine-tuning) are everyday edits of a
    reference model (vit.py:137-138 just calls `self.mlp_head(x)`): the drop-in calls whatever sits there instead of assuming the Linear
    its constructor built."""
    cfg = dict(image_size=32, patch_size=8, num_classes=7, dim=32, depth=2, heads=2, dim_head=16, mlp_dim=64)
    if kind == "vit":
        cfg["pool"] = "cls"
    params = make_params(kind, cfg, 79)
    img = make_images(cfg, 3, 1079)
    m = (ViT if kind == "vit" else SimpleViT)(**cfg)
    m.load_state_dict(params, strict=True)
    head_name = "mlp_head" if kind == "vit" else "linear_head"
    w, b = params[head_name + ".weight"], params[head_name + ".bias"]
    with KD.installed():
        logits = m(img)
        setattr(m, head_name, torch.nn.Identity())
        feats = m(img)
        assert tuple(feats.shape) == (3, cfg["dim"])
        assert rel(feats @ w.T + b, logits) <= 1e-5                    # the features ARE the input of the head that was there
        new_head = torch.nn.Sequential(torch.nn.LayerNorm(cfg["dim"]), torch.nn.Linear(cfg["dim"], 3))
        setattr(m, head_name, new_head)
        out = m(img)
        O.loss_fn(out).backward()
    assert tuple(out.shape) == (3, 3) and new_head[1].weight.grad is not None
    assert all(p.grad is not None for n, p in m.named_parameters() if p.numel())


class _LoRALinear(torch.nn.Linear):
    """what PEFT-style code puts in place of a Linear: a subclass whose forward adds a low-rank update"""

    def __init__(self, base: torch.nn.Linear, rank=4):
        super().__init__(base.in_features, base.out_features, bias=base.bias is not None)
        self.weight = base.weight
        if base.bias is not None:
            self.bias = base.bias
        g = torch.Generator().manual_seed(5)
        self.A = torch.nn.Parameter(0.1 * torch.randn(rank, base.in_features, generator=g))
        self.B = torch.nn.Parameter(0.1 * torch.randn(base.out_features, rank, generator=g))

    def forward(self, x):
        return torch.nn.functional.linear(x, self.weight + self.B @ self.A, self.bias)


def _count_fused(monkeypatch):
    from vit_pytorch_amd import engine as E
    calls = {"transformer": 0, "embed": 0, "head": 0}
    for name, fn in (("transformer", E.TransformerFn), ("embed", E.PatchEmbedFn), ("head", E.HeadFn)):
        orig = fn.apply

        def wrapped(*a, _orig=orig, _name=name, **kw):
            calls[_name] += 1
            return _orig(*a, **kw)
        monkeypatch.setattr(fn, "apply", staticmethod(wrapped))
    return calls


@pytest.mark.parametrize("kind", ["vit", "simple_vit"])
def test_model_surgery_takes_the_module_path_and_pristine_models_the_fused_one(kind, monkeypatch):
    """The fused stages read parameters and never call the modules -- legitimate only while the modules are exactly what the constructor
    built.  A pristine model runs the three fused stages (embedding, Transformer, head); after typical surgery (a LoRA subclass of Linear
    in place of to_qkv, a custom stem, an extra block) the affected stage runs op by op, the modules that are there get called, and the
    result is the one the edited model defines (the oracle on the merged weights)."""
    cfg = dict(image_size=32, patch_size=8, num_classes=7, dim=32, depth=2, heads=2, dim_head=16, mlp_dim=64)
    if kind == "vit":
        cfg["pool"] = "cls"
    params = make_params(kind, cfg, 81)
    img = make_images(cfg, 3, 1081)
    cls = ViT if kind == "vit" else SimpleViT
    calls = _count_fused(monkeypatch)
    with KD.installed():
        m = cls(**cfg); m.load_state_dict(params, strict=True)
        ref = m(img)
        assert calls == {"transformer": 1, "embed": 1, "head": 1}
        # 1. LoRA on the first layer's to_qkv
        attn = m.transformer.layers[0][0]
        lora = _LoRALinear(attn.to_qkv)
        attn.to_qkv = lora
        out = m(img)
        assert calls["transformer"] == 1 and calls["embed"] == 2           # the stack ran through its modules
        merged = dict(params); merged["transformer.layers.0.0.to_qkv.weight"] = (lora.weight + lora.B @ lora.A).detach()
        want, _ = O.run_fwd_bwd(kind, cfg, merged, img, torch.float32)
        assert rel(out, want) <= 2e-5 and rel(out, ref) > 1e-3
        O.loss_fn(out).backward()
        assert lora.A.grad is not None and lora.B.grad is not None and lora.A.grad.abs().sum() > 0
        # 2. a custom stem in place of to_patch_embedding (same output shape): the embedding stage is simply called
        m2 = cls(**cfg); m2.load_state_dict(params, strict=True)
        n_patches = (32 // 8) ** 2

        class Stem(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.proj = torch.nn.Linear(3 * 8 * 8, cfg["dim"])

            def forward(self, x):
                b = x.shape[0]
                return self.proj(x.reshape(b, 3, 4, 8, 4, 8).permute(0, 2, 4, 3, 5, 1).reshape(b, n_patches, -1))
        before = dict(calls)
        m2.to_patch_embedding = Stem()
        y = m2(img)
        assert tuple(y.shape) == tuple(ref.shape) and calls["embed"] == before["embed"] and calls["transformer"] == before["transformer"] + 1
        # 3. a foreign block appended to the stack
        m3 = cls(**cfg); m3.load_state_dict(params, strict=True)
        m3.transformer.layers.append(torch.nn.ModuleList([torch.nn.Identity(), torch.nn.Identity()]))
        before = dict(calls)
        y3 = m3(img)
        assert calls["transformer"] == before["transformer"] and torch.isfinite(y3).all()


@pytest.mark.parametrize("flags", [{"VITK_RECOMPUTE": "1"}, {"VITK_FWD_STREAM": "f32", "VITK_GRAD_STREAM": "f32"}, {"VITK_FWD_STREAM": "16"},
                                   {"VITK_GELU_DG": "0"}, {"VITK_GELU_DG": "16"}, {"VITK_RECOMPUTE": "1", "VITK_GRAD_STREAM": "f32"}])
@pytest.mark.parametrize("seed", [2, 4, 9, 19, 21, 29])
def test_fuzz_draw_under_engine_switches_bf16_host_logic(seed, flags, monkeypatch):
    """The engine's switches (recompute, stream dtypes, FeedForward pair) change WHICH tensors the host code allocates, saves and hands to
    which kernel; with the doubles in bfloat16 every combination must give the oracle's result to 16-bit accuracy on the large-M draws
    (tests/test_fuzz_gpu.py runs the same matrix on the real kernels)."""
    for k, v in flags.items():
        monkeypatch.setenv(k, v)
    kind, cfg, batch = F.draw(seed)
    params = make_params(kind, cfg, 50 + seed)
    img = make_images(cfg, batch, 1050 + seed)
    ref_out, ref_g = O.run_fwd_bwd(kind, cfg, params, img, torch.float32)
    m = (ViT if kind == "vit" else SimpleViT)(**cfg)
    m.load_state_dict(params, strict=True)
    m = m.to(torch.bfloat16)
    with KD.installed():
        out = m(img.to(torch.bfloat16))
        O.loss_fn(out).backward()
    keys = [k for k in params if params[k].numel()]
    cat = lambda d: torch.cat([d[k].detach().float().flatten() for k in keys])
    e, g = rel(out.float(), ref_out), rel(cat({k: p.grad for k, p in m.named_parameters()}), cat(ref_g))
    assert e <= 3e-2 and g <= 6e-2, (flags, kind, cfg, batch, e, g)


@pytest.mark.parametrize("seed", [4, 7, 29, 42, 46, 70, 87, 92, 117, 154, 160, 161, 177, 190])
def test_fuzz_draw_fp8_host_logic(seed):
    """enable_fp8 on large-M ViT draws through the doubles: three steps (record, first fp8 step, steady state of the delayed scales) --
    which GEMM of which layer runs on which operand format, which scale slot it reads, what is kept for the backward is all host logic.
    Tolerance: the fp8 one (a double quantises exactly like the kernels: 3-bit / 2-bit mantissas).  Seeds 4 / 29: shapes the fp8 GEMMs do
    not serve (mlp width % 64, too few rows for the 256-row kernel) -- the model then stays on the 16-bit kernels; the others are the draws
    below 200 on which ops.fp8_gemm_ok engages."""
    from vit_pytorch_amd.fp8 import enable_fp8
    kind, cfg, batch = F.draw(seed)
    assert kind == "vit"
    params = make_params(kind, cfg, 50 + seed)
    img = make_images(cfg, batch, 1050 + seed)
    ref_out, ref_g = O.run_fwd_bwd(kind, cfg, params, img, torch.float32)
    m = ViT(**cfg)
    m.load_state_dict(params, strict=True)
    m = m.to(torch.bfloat16)
    enable_fp8(m)
    with KD.installed() as calls:
        for _ in range(3):
            del calls[:]
            m.zero_grad(set_to_none=True)
            out = m(img.to(torch.bfloat16))
            O.loss_fn(out).backward()
        names = [c[0] for c in calls]
    keys = [k for k in params if params[k].numel()]
    cat = lambda d: torch.cat([d[k].detach().float().flatten() for k in keys])
    e, g = rel(out.float(), ref_out), rel(cat({k: p.grad for k, p in m.named_parameters()}), cat(ref_g))
    st = m.transformer._fp8
    from vit_pytorch_amd import ops
    n_tok = (cfg["image_size"][0] // cfg["patch_size"][0]) * (cfg["image_size"][1] // cfg["patch_size"][1]) + 1
    M_, D_, I_, F_ = batch * n_tok, cfg["dim"], cfg["heads"] * cfg["dim_head"], cfg["mlp_dim"]
    engaged = ops.fp8_gemm_ok(M_, D_, I_, F_)
    engaged_bwd = engaged and ops.fp8_out_ok(M_, D_, I_) and ops.fp8_bwd_ok(M_, D_, I_, F_)
    # shapes the fp8 GEMMs do not serve stay on the 16-bit kernels, silently -- the whole stack, or (inner width off the e5m2 dX / dW kernels'
    # multiples: dim_head 16 / 24 / 48 / 96 with few heads) only its backward
    assert st.ready == engaged and st.bwd_ready == engaged_bwd
    assert engaged == any(n.startswith("gemm_nt_fp8") for n in names)
    print(f"fp8 host logic draw {seed}: logits {e:.2e} grads {g:.2e}; fp8 GEMM calls {sum(n.startswith('gemm_nt_fp8') or n == 'gemm_tn_fp8' for n in names)}")
    assert e <= 6e-2 and g <= 1.5e-1, (cfg, batch, e, g)


def test_navit_surgery_takes_the_module_path():
    """NaViT in bfloat16 runs the fused packed stack only on blocks that are exactly what its constructor builds (na_vit.py:96-169); a LoRA
    subclass in place of `to_q` takes the op-by-op path, is called, changes the result and trains."""
    from oracle.params import NAVIT_CASES
    case = NAVIT_CASES["navit_two_packs"]
    params = make_navit_params(case["cfg"], case["seed"])
    imgs = make_navit_images(case["cfg"], case["sizes"], case["seed"] + 1000)
    m = NaViT(**case["cfg"])
    m.load_state_dict(params, strict=True)
    m = m.to(torch.bfloat16).eval()
    probe = torch.empty(4, case["cfg"]["dim"], dtype=torch.bfloat16)
    assert m.transformer._fusable(probe)
    x = [[im.to(torch.bfloat16) for im in g] for g in imgs]
    with KD.installed():
        base = m(x)
        attn = m.transformer.layers[0][0]
        lora = _LoRALinear(attn.to_q.float()).to(torch.bfloat16)
        attn.to_q = lora
        assert not m.transformer._fusable(probe)
        out = m(x)
        O.loss_fn(out).backward()
    assert rel(out.float(), base.float()) > 1e-3
    assert lora.A.grad is not None and lora.A.grad.float().abs().sum() > 0 and lora.B.grad is not None
