"""GPU: several optimizer steps in a row -- the derived weight copies (K-blocked packs for the persistent NT GEMM, W^T for dX, the NaViT
q|kv concatenation, e4m3 weights and their scales under enable_fp8) are cached between forwards and keyed on the parameter's
version counter (_epoch.py); an optimizer's in-place update must invalidate every one of them.  A stale copy gives a PLAUSIBLE
wrong result (last step's weights), so the check is exact: the same training run with VITK_WEIGHT_CACHE=0 (every use re-derives its
copies) must produce bit-identical logits and parameters.  torch.optim (foreach and fused flavours) is what the reference's
training step uses (train_vit_decorr.py:68-70,110)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.params import make_images, make_params  # noqa: E402
from vit_pytorch_amd import ViT  # noqa: E402

DEV = "cuda"
CFG = dict(image_size=112, patch_size=8, num_classes=10, dim=256, depth=2, heads=4, dim_head=64, mlp_dim=1024)     # N = 197, batch 8: M = 1576


def _train(opt_name, fp8, steps=4):
    params = make_params("vit", CFG, 11)
    m = ViT(**CFG)
    m.load_state_dict(params)
    m = m.to(DEV, dtype=torch.bfloat16)
    if fp8:
        from vit_pytorch_amd.fp8 import enable_fp8
        enable_fp8(m)
    opt = {"sgd": lambda: torch.optim.SGD(m.parameters(), lr=0.05, momentum=0.9),
           "adamw_foreach": lambda: torch.optim.AdamW(m.parameters(), lr=3e-4, foreach=True),
           "adamw_fused": lambda: torch.optim.AdamW(m.parameters(), lr=3e-4, fused=True)}[opt_name]()
    losses = []
    for it in range(steps):
        x = make_images(CFG, 8, 2000 + it).to(DEV, dtype=torch.bfloat16)
        opt.zero_grad(set_to_none=True)
        loss = m(x).float().square().mean()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    with torch.no_grad():
        final = m(make_images(CFG, 8, 2999).to(DEV, dtype=torch.bfloat16))
    return losses, final, [p.detach().clone() for p in m.parameters()]


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("opt_name", ["sgd", "adamw_foreach", "adamw_fused"])
def test_weight_caches_follow_the_optimizer(opt_name, fp8, monkeypatch):
    monkeypatch.delenv("VITK_NTP_EPIS", raising=False)
    monkeypatch.delenv("VITK_PACK_W", raising=False)
    monkeypatch.delenv("VITK_WEIGHT_CACHE", raising=False)
    losses, final, ps = _train(opt_name, fp8)
    monkeypatch.setenv("VITK_WEIGHT_CACHE", "0")
    losses0, final0, ps0 = _train(opt_name, fp8)
    assert losses == losses0, (losses, losses0)
    assert torch.equal(final, final0)
    assert all(torch.equal(a, b) for a, b in zip(ps, ps0))
    assert losses[-1] < losses[0]                     # and it trains
