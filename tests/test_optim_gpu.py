"""GPU: the fused flat-buffer Adam / AdamW (vit_pytorch_amd/optim.py) against torch.optim step by step.
torch.optim.Adam is what the reference's training step uses (train_vit_decorr.py:68-70,110)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from vit_pytorch_amd import ViT, kernels as K  # noqa: E402
from vit_pytorch_amd.optim import Adam, AdamW  # noqa: E402
from vit_pytorch_amd.parallel import DataParallel  # noqa: E402

DEV = "cuda"
CFG = dict(image_size=32, patch_size=8, num_classes=10, dim=64, depth=2, heads=2, dim_head=32, mlp_dim=128)


def _model(dtype):
    torch.manual_seed(0)
    return ViT(**CFG).to(DEV, dtype=dtype)


@pytest.mark.parametrize("cls,kw", [(Adam, dict(weight_decay=0.0)), (Adam, dict(weight_decay=0.1)), (AdamW, dict(weight_decay=0.1))])
def test_fused_adam_matches_torch_f32(cls, kw):
    m = _model(torch.float32)
    dp = DataParallel(m)
    names = [n for n, _ in m.named_parameters()]
    ref = {n: p.detach().clone().requires_grad_(True) for n, p in m.named_parameters()}
    topt = (torch.optim.AdamW if cls is AdamW else torch.optim.Adam)(list(ref.values()), lr=1e-2, betas=(0.8, 0.95), eps=1e-6, **kw)
    opt = cls(dp, lr=1e-2, betas=(0.8, 0.95), eps=1e-6, **kw)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    for k, v in m.state_dict().items():                       # re-homing must not change any value
        assert torch.equal(v, sd0[k])
    g = torch.Generator(device=DEV).manual_seed(3)
    for it in range(4):
        x = torch.randn(6, 3, 32, 32, device=DEV, generator=g)
        loss = dp(x).float().square().mean()
        dp.backward(loss)
        for n, p in m.named_parameters():                     # same gradients into the torch optimizer
            ref[n].grad = p.grad.detach().clone()
        opt.step()
        topt.step()
        for n, p in m.named_parameters():
            assert torch.allclose(p.detach(), ref[n].detach(), rtol=2e-5, atol=2e-6), (it, n)
    assert len(names) == len(ref)


def test_fused_adam_bf16_master_weights_and_state_dict():
    m = _model(torch.bfloat16)
    dp = DataParallel(m)
    opt = Adam(dp, lr=3e-3)
    master0 = opt.master.clone()
    ref = master0.clone().requires_grad_(True)                  # f32 torch Adam on the flat master copy
    topt = torch.optim.Adam([ref], lr=3e-3)
    g = torch.Generator(device=DEV).manual_seed(4)
    for it in range(3):
        x = torch.randn(4, 3, 32, 32, device=DEV, generator=g).to(torch.bfloat16)
        dp.backward(dp(x).float().square().mean())
        ref.grad = dp.sink.flat.float()
        opt.step(); topt.step()
        assert torch.allclose(opt.master, ref.detach(), rtol=2e-5, atol=1e-7)
        assert torch.equal(opt.flat_p, opt.master.to(torch.bfloat16))          # parameters = rounded master copy
    # parameters are views of the flat buffer and the engine still finds their gradient slots
    assert all(p.data_ptr() >= opt.flat_p.data_ptr() for p in m.parameters() if p.numel())
    sd = opt.state_dict()
    opt2_model = _model(torch.bfloat16)
    dp2 = DataParallel(opt2_model)
    opt2 = Adam(dp2, lr=1.0)
    opt2.load_state_dict(sd)
    assert opt2.t == 3 and opt2.lr == 3e-3 and torch.equal(opt2.exp_avg, opt.exp_avg) and torch.equal(opt2.master, opt.master)


def test_adam_kernel_tail_and_errors():
    n = 1027                                                    # not a multiple of 4: scalar tail
    p = torch.randn(n, device=DEV); g = torch.randn(n, device=DEV)
    m = torch.zeros(n, device=DEV); v = torch.zeros(n, device=DEV)
    pr = p.clone().requires_grad_(True); pr.grad = g.clone()
    K.adam_step(p, g, m, v, None, n, 1e-2, 0.9, 0.999, 1e-8, 0.0, False, 1)
    t = torch.optim.Adam([pr], lr=1e-2); t.step()
    assert torch.allclose(p, pr.detach(), rtol=1e-5, atol=1e-7)
    with pytest.raises(Exception):
        K.adam_step(p, g, m, v, None, n, 1e-2, 0.9, 0.999, 1e-8, 0.0, False, 0)    # step counts from 1
