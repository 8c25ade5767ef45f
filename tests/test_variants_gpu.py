"""GPU: the sibling variants of SURVEY 8f item 4 -- SimpleViT with q/k normalisation, SimpleViT with register tokens, ViT with
patch dropout -- against goldens produced by the reference's own modules (oracle/make_golden.py::main_variants; eval mode, f32).
f32 mode is held to the north star's 1e-3 (logits and every parameter gradient); bf16 to 3e-2 of the f32 reference."""
import importlib
import json
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.params import VARIANT_CASES, make_images, make_params_for  # noqa: E402
from oracle.vit_oracle import loss_fn  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda"


def rel(a, b):
    a = a.detach().double().flatten().cpu(); b = b.detach().double().flatten().cpu()
    n = b.norm().item()
    return (a - b).norm().item() / (n if n > 0 else 1.0)


def build(name, dtype):
    case = VARIANT_CASES[name]
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    mod = importlib.import_module("vit_pytorch_amd." + case["module"])
    m = getattr(mod, case["cls"])(**case["cfg"])
    ref_shapes = OrderedDict((k, tuple(v)) for k, v in json.loads(bytes(gold["state_dict_shapes"]).decode()))
    mine = OrderedDict((k, tuple(v.shape)) for k, v in m.state_dict().items())
    assert mine == ref_shapes, "state_dict contract differs from the reference's"      # keys, shapes AND order
    m.load_state_dict(make_params_for(ref_shapes, case["seed"]), strict=True)
    img = make_images(case["cfg"], case["batch"], case["seed"] + 1000)
    return m.to(DEV, dtype=dtype).eval(), img.to(DEV, dtype=dtype), gold


@pytest.mark.parametrize("name", list(VARIANT_CASES))
def test_variant_f32_matches_reference_golden(name):
    m, img, gold = build(name, torch.float32)
    out = m(img)
    loss_fn(out).backward()
    assert tuple(out.shape) == gold["logits"].shape
    e = rel(out, torch.from_numpy(gold["logits"]))
    assert e <= 1e-3, e
    for k, p in m.named_parameters():
        g_ref = torch.from_numpy(gold["grad::" + k])
        assert p.grad is not None, k
        eg = rel(p.grad, g_ref)
        assert eg <= 1e-3, (k, eg)


@pytest.mark.parametrize("name", list(VARIANT_CASES))
def test_variant_bf16_close_to_reference_golden(name):
    m, img, gold = build(name, torch.bfloat16)
    out = m(img)
    loss_fn(out).backward()
    e = rel(out, torch.from_numpy(gold["logits"]))
    keys = [k for k, _ in m.named_parameters()]
    gm = torch.cat([p.grad.float().flatten().cpu() for _, p in m.named_parameters()])
    gr = torch.cat([torch.from_numpy(gold["grad::" + k]).flatten() for k in keys])
    g = rel(gm, gr)
    print(f"{name} bf16: logits {e:.2e} grads {g:.2e}")
    assert e <= 3e-2 and g <= 3e-2, (e, g)


def test_patch_dropout_training_mode_keeps_a_subset():
    from vit_pytorch_amd.vit_with_patch_dropout import ViT
    torch.manual_seed(0)
    m = ViT(image_size=32, patch_size=8, num_classes=5, dim=64, depth=1, heads=2, mlp_dim=64, patch_dropout=0.5).to(DEV).train()
    seen = []
    h = m.transformer.register_forward_hook(lambda mod, i, o: seen.append(o.shape))
    x = torch.randn(4, 3, 32, 32, device=DEV)
    out = m(x)
    out.square().mean().backward()
    h.remove()
    assert tuple(seen[0]) == (4, 1 + 8, 64)                       # cls + max(1, int(16 * 0.5)) kept patches
    gp = m.pos_embedding.grad
    assert gp is not None and torch.isfinite(gp).all()
    rows_touched = (gp.abs().sum(-1) > 0).sum().item()
    assert 8 <= rows_touched <= 16                                # only rows some image kept receive gradient
    m.eval()
    seen.clear()
    h = m.transformer.register_forward_hook(lambda mod, i, o: seen.append(o.shape))
    m(x)
    h.remove()
    assert tuple(seen[0]) == (4, 17, 64)                          # eval: every patch


def test_register_tokens_do_not_change_the_pooled_token_count():
    from vit_pytorch_amd.simple_vit_with_register_tokens import SimpleViT
    m = SimpleViT(image_size=32, patch_size=8, num_classes=5, dim=64, depth=1, heads=2, mlp_dim=64, num_register_tokens=3).to(DEV)
    seen = []
    h = m.transformer.register_forward_hook(lambda mod, i, o: seen.append(o.shape))
    out = m(torch.randn(2, 3, 32, 32, device=DEV))
    h.remove()
    assert tuple(seen[0]) == (2, 16 + 3, 64) and tuple(out.shape) == (2, 5)
    out.sum().backward()
    assert m.register_tokens.grad is not None and m.register_tokens.grad.abs().sum().item() > 0


def test_register_tokens_follow_the_patch_tokens_like_the_reference_packs_them():
    """simple_vit_with_register_tokens.py:113-115: `pack([x, r], 'b * d')` -- the registers come AFTER the patch tokens.  A forward
    pre-hook on the transformer (what Extractor-style consumers attach) must see that order: the last R rows of every image are the
    register parameters themselves, the first rows the position-encoded patch embeddings (rounds 4-5 had them in front)."""
    from vit_pytorch_amd.simple_vit_with_register_tokens import SimpleViT
    R = 3
    m = SimpleViT(image_size=32, patch_size=8, num_classes=5, dim=64, depth=1, heads=2, mlp_dim=64, num_register_tokens=R).to(DEV)
    seen = []
    h = m.transformer.register_forward_pre_hook(lambda mod, args: seen.append(args[0].detach().clone()))
    img = torch.randn(2, 3, 32, 32, device=DEV)
    out = m(img)
    h.remove()
    x = seen[0]
    assert tuple(x.shape) == (2, 16 + R, 64)
    assert torch.equal(x[:, 16:], m.register_tokens.detach().expand(2, R, 64))
    patches = m.to_patch_embedding(img) + m.pos_embedding.to(DEV)
    assert torch.allclose(x[:, :16], patches, rtol=1e-5, atol=1e-6)
    out.square().mean().backward()
    assert m.register_tokens.grad is not None and m.register_tokens.grad.abs().sum().item() > 0
