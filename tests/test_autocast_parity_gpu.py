"""GPU: the drop-in's AUTOCAST route against the reference's own autocast run (round 6).

`with torch.autocast("cuda", dtype=torch.bfloat16): model(x)` on float32 master parameters is what accelerate's mixed precision runs
around the reference (train_vit_decorr.py:74-78, vit.py:118-138): Linear / matmul in bfloat16, LayerNorm / softmax / the residual stream
in float32.  tests/golden/<case>__autocast.npz holds logits and gradient samples of the UNMODIFIED reference executed that way on the CPU
(oracle/make_golden.py::main_autocast), next to the float32 golden of the same parameters and images.

Stated tolerance: the drop-in's distance from the reference's float32 run is at most 1.5 x the reference-autocast's own distance from
it + 1e-3, for the logits and for the concatenated gradient samples -- the rule the pure-bf16 mode is held to against the reference's
pure-bf16 run, now for the contract mixed-precision users actually run."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import vit_oracle as O  # noqa: E402
from oracle.params import WIDE_CASES, make_images, make_params, sample_index  # noqa: E402
from vit_pytorch_amd import SimpleViT, ViT  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = [n[:-len("__autocast.npz")] for n in sorted(os.listdir(GOLD)) if n.endswith("__autocast.npz")]


def rel(a, b):
    a = a.detach().double().flatten().cpu(); b = b.detach().double().flatten().cpu()
    n = b.norm().item()
    return (a - b).norm().item() / (n if n > 0 else 1.0)


def test_the_autocast_goldens_exist():
    assert "vit_b16_width" in CASES and "vit_b16_full" in CASES, CASES


@pytest.mark.parametrize("name", CASES)
def test_autocast_route_vs_reference_autocast(name):
    case = WIDE_CASES[name]
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    gold_ac = np.load(os.path.join(GOLD, name + "__autocast.npz"))
    params = make_params(case["kind"], case["cfg"], case["seed"])
    img = make_images(case["cfg"], case["batch"], case["seed"] + 1000)
    m = (ViT if case["kind"] == "vit" else SimpleViT)(**case["cfg"])
    m.load_state_dict(params, strict=True)
    m = m.to("cuda")                                  # float32 master parameters
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = m(img.to("cuda"))
    assert out.dtype == torch.bfloat16                # like the reference's last Linear under autocast
    O.loss_fn(out.float()).backward()
    ref_logits = torch.from_numpy(gold["logits"])
    ac_logits = torch.from_numpy(gold_ac["autocast::logits"])
    mine, ref, ac = [], [], []
    for k, p in m.named_parameters():
        if not p.numel():
            continue
        assert p.grad is not None and p.grad.dtype == torch.float32, k
        g = p.grad.detach().float().flatten().cpu()
        idx = torch.from_numpy(sample_index(g.numel(), case.get("sample", 4096)))
        mine.append(g[idx]); ref.append(torch.from_numpy(gold["gsample::" + k]).float()); ac.append(torch.from_numpy(gold_ac["autocast::gsample::" + k]).float())
    e, e_ac = rel(out.float(), ref_logits), rel(ac_logits, ref_logits)
    g, g_ac = rel(torch.cat(mine), torch.cat(ref)), rel(torch.cat(ac), torch.cat(ref))
    print(f"{name} under autocast(bf16): logits {e:.2e} (reference-autocast {e_ac:.2e}) gradient samples {g:.2e} (reference-autocast {g_ac:.2e})")
    assert e <= 1.5 * e_ac + 1e-3, (e, e_ac)
    assert g <= 1.5 * g_ac + 1e-3, (g, g_ac)
