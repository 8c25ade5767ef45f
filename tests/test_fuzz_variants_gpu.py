"""GPU: the sibling variants and the dropout paths on seeded random configurations.

No CPU oracle restates the variants, so these are CONSISTENCY properties the reference's modules have by construction -- two
different kernel routes of this tree must agree on them:
  * a variant in f32 (materialising / coverage kernels) and in bf16 (flash kernels, fused stages) computes the same function:
    logits and the concatenated gradient within 3e-2 relative L2 (bf16 round-off of a 1-2 layer model), both finite;
  * eval() of a model built with dropout > 0 equals the same weights in a model built with dropout = 0 (nn.Dropout is the identity
    in eval: vit.py:22,24,42,48,60,114), bit for bit; train() with dropout is finite, differs between two calls, and keeps E[x].
The fixed-shape goldens of tests/test_variants_gpu.py pin the variants to the reference itself at dim_head 64."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.params import make_images, make_params  # noqa: E402
from oracle import vit_oracle as O  # noqa: E402
from vit_pytorch_amd import ViT  # noqa: E402
from vit_pytorch_amd import simple_vit_with_qk_norm as QK  # noqa: E402
from vit_pytorch_amd import simple_vit_with_register_tokens as RT  # noqa: E402
from vit_pytorch_amd import vit_with_patch_dropout as PD  # noqa: E402

DEV = "cuda"


def rel(a, b):
    a = a.detach().double().flatten().cpu(); b = b.detach().double().flatten().cpu()
    n = b.norm().item()
    return (a - b).norm().item() / (n if n > 0 else 1.0)


def _fwd_bwd(m, img, dtype):
    m = m.to(DEV, dtype=dtype)
    out = m(img.to(DEV, dtype=dtype))
    O.loss_fn(out).backward()
    g = torch.cat([p.grad.float().flatten() for p in m.parameters() if p.numel()])
    return out.float(), g


def _variant(seed):
    r = np.random.RandomState(300 + seed)
    which = ["qk_norm", "register", "patch_dropout"][seed % 3]
    p = int(r.choice([4, 8, 16]))
    g = int(r.choice([3, 5, 8, 14]))
    cfg = dict(image_size=g * p, patch_size=p, num_classes=int(r.choice([5, 1000])), dim=int(r.choice([64, 128, 256, 384])), depth=int(r.choice([1, 2])),
               heads=int(r.choice([1, 2, 4, 6])), mlp_dim=int(r.choice([128, 200, 512])), dim_head=int(r.choice([32, 48, 64, 80, 96] if which == "qk_norm" else [16, 32, 64, 80])))
    if which == "register":
        cfg["num_register_tokens"] = int(r.choice([1, 4, 7]))
    if which == "patch_dropout":
        cfg["patch_dropout"] = 0.25
    batch = int(r.choice([1, 3, 9, 24]))
    return which, cfg, batch


@pytest.mark.parametrize("seed", range(18))
def test_variant_f32_and_bf16_routes_agree(seed):
    which, cfg, batch = _variant(seed)
    mod = {"qk_norm": QK.SimpleViT, "register": RT.SimpleViT, "patch_dropout": PD.ViT}[which]
    torch.manual_seed(seed)
    m32 = mod(**cfg).eval()                     # eval: patch dropout keeps every patch (vit_with_patch_dropout.py:25-26)
    sd = {k: v.clone() for k, v in m32.state_dict().items()}
    img = torch.randn(batch, 3, cfg["image_size"], cfg["image_size"])
    o32, g32 = _fwd_bwd(m32, img, torch.float32)
    m16 = mod(**cfg).eval()
    m16.load_state_dict(sd, strict=True)
    o16, g16 = _fwd_bwd(m16, img, torch.bfloat16)
    assert torch.isfinite(o32).all() and torch.isfinite(o16).all() and torch.isfinite(g32).all() and torch.isfinite(g16).all(), (which, cfg, batch)
    e, g = rel(o16, o32), rel(g16, g32)
    print(f"variant draw {seed}: {which} {cfg} batch {batch}: bf16 vs f32 logits {e:.2e} grads {g:.2e}")
    assert e <= 3e-2 and g <= 3e-2, (which, cfg, batch, e, g)


@pytest.mark.parametrize("seed", range(10))
def test_dropout_models_eval_identity_and_train_statistics(seed):
    r = np.random.RandomState(500 + seed)
    p = int(r.choice([4, 8, 16]))
    g = int(r.choice([4, 7, 14]))
    cfg = dict(image_size=g * p, patch_size=p, num_classes=10, dim=int(r.choice([64, 128, 256, 768])), depth=int(r.choice([1, 2])),
               heads=int(r.choice([2, 4, 12])), dim_head=int(r.choice([32, 64, 64, 80])), mlp_dim=int(r.choice([128, 512, 1024])),
               pool=str(r.choice(["cls", "mean"])))
    batch = int(r.choice([2, 6, 16, 40]))           # 40 x 197 tokens: the fused-dropout epilogues (engine.dropout_fusable) take over
    params = make_params("vit", cfg, 700 + seed)
    img = make_images(cfg, batch, 1700 + seed).to(DEV, dtype=torch.bfloat16)
    plain = ViT(**cfg); plain.load_state_dict(params); plain = plain.to(DEV, dtype=torch.bfloat16).eval()
    drop = ViT(**cfg, dropout=0.2, emb_dropout=0.1); drop.load_state_dict(params); drop = drop.to(DEV, dtype=torch.bfloat16)
    with torch.no_grad():
        ref = plain(img)
        assert torch.equal(drop.eval()(img), ref), cfg
    drop.train()
    torch.manual_seed(seed)
    outs = []
    for _ in range(6):
        drop.zero_grad(set_to_none=True)
        out = drop(img)
        O.loss_fn(out).backward()
        assert torch.isfinite(out).all() and all(torch.isfinite(q.grad).all() for q in drop.parameters() if q.numel()), cfg
        outs.append(out.detach().float())
    assert not torch.equal(outs[0], outs[1]), "two training calls drew the same masks"
    # inverted dropout keeps expectations: the mean over calls stays near the dropout-free logits (loose: 6 draws through 1-2 layers)
    mean = torch.stack(outs).mean(0)
    assert rel(mean, ref.float()) < 0.8, (cfg, rel(mean, ref.float()))
