"""CPU: the complete drop-in models (patch embedding -> Transformer -> head: engine.PatchEmbedFn / TransformerFn / HeadFn) with the kernels
replaced by the test doubles (tests/_kernel_doubles.py), in bfloat16, against the goldens the REFERENCE produced in float32
(tests/golden/*.npz, oracle/make_golden.py).  Host logic only -- row maps, cls / positional handling, padding of odd patch widths,
pooling, which gradient lands in which parameter; the kernels are checked on the GPU (tests/test_parity_gpu.py runs the same cases
there).  Tolerances: float32 parameters -- round-off (logits 2e-5, every gradient tensor 1e-4); bfloat16 against the f32 outputs on these tiny
models -- logits 2e-2, every gradient tensor 6e-2 of its norm (measured 2e-3 .. 6e-3 and <= 2e-2)."""
import os

import numpy as np
import pytest
import torch

from oracle import vit_oracle as O
from oracle.params import CASES, make_images, make_params
from vit_pytorch_amd import SimpleViT, ViT

import _kernel_doubles as KD

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    a = a.detach().double().flatten(); b = b.detach().double().flatten()
    n = b.norm().item()
    return (a - b).norm().item() / (n if n > 0 else 1.0)


@pytest.mark.parametrize("name", list(CASES))
def test_dropin_models_against_reference_goldens_f32(name):
    """float32 parameters: the doubles compute in float32, so the engine's host logic is held to the reference's outputs to round-off."""
    case = CASES[name]
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    params = make_params(case["kind"], case["cfg"], case["seed"])
    img = make_images(case["cfg"], case["batch"], case["seed"] + 1000, case.get("image"))
    m = (ViT if case["kind"] == "vit" else SimpleViT)(**case["cfg"])
    m.load_state_dict(params, strict=True)
    with KD.installed():
        out = m(img)
        O.loss_fn(out).backward()
    e = rel(out, torch.from_numpy(gold["logits"]))
    assert e < 2e-5, e
    for k, p in m.named_parameters():
        g_ref = torch.from_numpy(gold["grad::" + k])
        if g_ref.numel():
            assert rel(p.grad, g_ref) < 1e-4, (k, rel(p.grad, g_ref))


@pytest.mark.parametrize("name", list(CASES))
def test_dropin_models_against_reference_goldens_bf16(name):
    case = CASES[name]
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    params = make_params(case["kind"], case["cfg"], case["seed"])
    img = make_images(case["cfg"], case["batch"], case["seed"] + 1000, case.get("image"))
    cls = ViT if case["kind"] == "vit" else SimpleViT
    m = cls(**case["cfg"])
    m.load_state_dict(params, strict=True)
    m = m.to(torch.bfloat16)
    with KD.installed():
        out = m(img.to(torch.bfloat16))
        O.loss_fn(out).backward()
    assert tuple(out.shape) == gold["logits"].shape
    e = rel(out.float(), torch.from_numpy(gold["logits"]))
    assert e < 2e-2, e
    worst = 0.0
    for k, p in m.named_parameters():
        g_ref = torch.from_numpy(gold["grad::" + k])
        if g_ref.numel() == 0:
            assert p.grad is None or p.grad.numel() == 0
            continue
        assert p.grad is not None, k
        worst = max(worst, rel(p.grad.float(), g_ref))
        assert rel(p.grad.float(), g_ref) < 6e-2, (k, rel(p.grad.float(), g_ref))
    print(f"{name}: logits {e:.2e}, worst gradient tensor {worst:.2e}")
