"""CPU, build container only (skipped where /root/reference is absent, e.g. on the GPU box): the oracle against the REFERENCE ITSELF on
the seeded random configurations the GPU fuzz tests use (tests/test_fuzz_gpu.py).  The committed goldens pin the oracle on fixed
cases; the fuzz leans on it for rectangular patches, 1 / 4 channels, mean pooling, free dim_head, ragged NaViT packs -- so those exact
draws are executed through the unmodified reference modules here (oracle/make_golden.run_reference loads them by file path):
logits within 1e-5 and every gradient within 1e-4 relative L2 in float32, and the error of the oracle's pure-bf16 run (the yardstick
of the fuzz's 1.5x gate) of the same size as the error of the reference's own bf16 run (factor 3 either way: at these model sizes a
few hundred roundings decide either figure; the BASELINE-width goldens carry the reference's own bf16 numbers instead)."""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/vit_pytorch"), reason="the reference is only present in the build container")

import test_fuzz_gpu as F  # noqa: E402  (the draws; importing it needs no GPU)
from oracle import make_golden as G  # noqa: E402
from oracle import navit_oracle as NO  # noqa: E402
from oracle import vit_oracle as O  # noqa: E402
from oracle.params import make_images, make_navit_images, make_navit_params, make_params  # noqa: E402


def rel(a, b):
    a = a.detach().double().flatten(); b = b.detach().double().flatten()
    n = b.norm().item()
    return (a - b).norm().item() / (n if n > 0 else 1.0)


@pytest.mark.parametrize("seed", range(F.N_DRAWS))
def test_oracle_equals_reference_on_fuzz_draw(seed):
    kind, cfg, batch = F.draw(seed)
    params = make_params(kind, cfg, 50 + seed)
    img = make_images(cfg, batch, 1050 + seed)
    out, grads = O.run_fwd_bwd(kind, cfg, params, img, torch.float32)
    ref_out, _, ref_g = G.run_reference(kind, cfg, params, img)
    assert tuple(out.shape) == tuple(ref_out.shape)
    assert rel(out, ref_out) <= 1e-5, (kind, cfg, batch)
    for k, g in ref_g.items():
        if g.numel():
            assert rel(grads[k], g) <= 1e-4, (k, kind, cfg, batch, rel(grads[k], g))
    if seed % 4 == 0:       # the bf16 yardstick on a quarter of the draws (the reference in bf16 on the CPU is slow)
        o16, g16 = O.run_fwd_bwd(kind, cfg, params, img, torch.bfloat16)
        r16, _, rg16 = G.run_reference(kind, cfg, params, img, torch.bfloat16)
        e_o, e_r = rel(o16, ref_out), rel(r16, ref_out)
        assert e_o <= 3 * e_r + 1e-3 and e_r <= 3 * e_o + 1e-3, (e_o, e_r)      # same size (small models: a few hundred roundings decide either figure)


@pytest.mark.parametrize("seed", range(8))
def test_navit_oracle_equals_reference_on_fuzz_draw(seed):
    cfg, packs = F.draw_navit(seed)
    params = make_navit_params(cfg, 90 + seed)
    images = make_navit_images(cfg, packs, 1090 + seed)
    out, grads = NO.run_fwd_bwd(cfg, params, images, torch.float32)
    model = G.load_ref("na_vit").NaViT(**cfg)
    model.load_state_dict(params, strict=True)
    model.eval()
    ref = model(images)
    O.loss_fn(ref).backward()
    assert rel(out, ref) <= 1e-5, (cfg, packs)
    for k, p in model.named_parameters():
        if p.numel():
            assert rel(grads[k], p.grad) <= 1e-4, (k, cfg, packs, rel(grads[k], p.grad))
