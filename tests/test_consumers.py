"""Consumers of the encoder's internals: the reference's Recorder (recorder.py:26-29), Extractor
(extractor.py:50-57) and Dino's NetWrapper (dino.py:134-151) wired around the drop-in modules.

CPU part (runs where /root/reference exists, i.e. in the build container): the reference's OWN wrapper classes are imported with
`vit_pytorch.vit` aliased to `vit_pytorch_amd.vit` and must find their hook points -- every `Attention.attend`, the `transformer`
attribute, `children()[-2]` = `to_latent`.  GPU part: the same wiring restated (the reference does not travel to the GPU box),
run end to end against the un-hooked fused path."""
import importlib.util
import os
import sys
import types

import pytest
import torch

REF = "/root/reference/vit_pytorch"
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists in the build container only")


def _load_ref_module(name):
    """Import /root/reference/vit_pytorch/<name>.py with `vit_pytorch.vit` resolving to the drop-in and torchvision stubbed."""
    sys.dont_write_bytecode = True
    import vit_pytorch_amd.vit as mine
    pkg = types.ModuleType("vit_pytorch"); pkg.__path__ = []
    saved = {k: sys.modules.get(k) for k in ("vit_pytorch", "vit_pytorch.vit", "torchvision", "torchvision.transforms")}
    sys.modules["vit_pytorch"] = pkg
    sys.modules["vit_pytorch.vit"] = mine
    tv = types.ModuleType("torchvision"); tvt = types.ModuleType("torchvision.transforms")
    tvt.__getattr__ = lambda attr: (lambda *a, **k: torch.nn.Identity())       # dino.py only builds augmentation pipelines from it
    tv.transforms = tvt
    sys.modules.setdefault("torchvision", tv); sys.modules.setdefault("torchvision.transforms", tvt)
    try:
        spec = importlib.util.spec_from_file_location(f"_ref_consumer_{name}", os.path.join(REF, f"{name}.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def _tiny_vit():
    from vit_pytorch_amd import ViT
    return ViT(image_size=32, patch_size=8, num_classes=5, dim=32, depth=3, heads=2, dim_head=16, mlp_dim=64)


@needs_ref
def test_reference_recorder_finds_every_attend_module():
    rec_mod = _load_ref_module("recorder")
    v = _tiny_vit()
    rec = rec_mod.Recorder(v)
    rec._register_hook()                                  # recorder.py:26-31
    assert rec.hook_registered and len(rec.hooks) == 3
    from vit_pytorch_amd.vit import Attention
    attends = [m.attend for m in v.transformer.modules() if isinstance(m, Attention)]
    assert len(attends) == 3 and all(len(a._forward_hooks) == 1 for a in attends)
    assert rec.eject() is v and all(len(a._forward_hooks) == 0 for a in attends)


@needs_ref
def test_reference_extractor_and_netwrapper_find_their_layers():
    ext_mod = _load_ref_module("extractor")
    v = _tiny_vit()
    ext = ext_mod.Extractor(v)
    ext._register_hook()                                  # extractor.py:50-59: the `transformer` attribute
    assert len(v.transformer._forward_hooks) == 1
    ext.eject()
    dino = _load_ref_module("dino")
    w = dino.NetWrapper(v, output_dim=16, projection_hidden_size=32, projection_num_layers=2, layer=-2)
    assert w._find_layer() is v.to_latent                 # dino.py:138-140: children()[-2]
    from vit_pytorch_amd import SimpleViT
    s = SimpleViT(image_size=32, patch_size=8, num_classes=5, dim=32, depth=1, heads=2, dim_head=16, mlp_dim=64)
    assert dino.NetWrapper(s, 16, 32, 2, layer=-2)._find_layer() is s.to_latent


# ---- GPU: the same wiring, end to end ------------------------------------------------------------------------------------
gpu = pytest.mark.gpu


def _rel(a, b):
    a = a.detach().double().flatten().cpu(); b = b.detach().double().flatten().cpu()
    return ((a - b).norm() / b.norm()).item()


@gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_recorder_wiring_end_to_end(dtype):
    from vit_pytorch_amd.vit import Attention
    torch.manual_seed(0)
    v = _tiny_vit().to("cuda", dtype=dtype).eval()
    img = torch.randn(2, 3, 32, 32, device="cuda").to(dtype)
    plain = v(img)
    recordings = []
    hooks = [m.attend.register_forward_hook(lambda _m, _i, o: recordings.append(o.clone().detach()))
             for m in v.transformer.modules() if isinstance(m, Attention)]
    pred = v(img)
    attns = torch.stack(recordings, dim=1)                # recorder.py:57-59
    for h in hooks:
        h.remove()
    assert tuple(attns.shape) == (2, 3, 2, 17, 17)        # (batch, depth, heads, n, n)
    assert torch.allclose(attns.float().sum(-1), torch.ones(2, 3, 2, 17, device="cuda"), atol=2e-2 if dtype == torch.bfloat16 else 1e-5)
    assert _rel(pred, plain) < (3e-2 if dtype == torch.bfloat16 else 1e-5)


@gpu
def test_extractor_and_netwrapper_wiring_end_to_end():
    torch.manual_seed(0)
    v = _tiny_vit().to("cuda").eval()
    img = torch.randn(2, 3, 32, 32, device="cuda")
    plain = v(img)
    got = {}
    h1 = v.transformer.register_forward_hook(lambda _m, _i, o: got.__setitem__("latents", o.clone().detach()))     # extractor.py:46-48
    h2 = [*v.children()][-2].register_forward_hook(lambda _m, i, o: got.__setitem__("hidden", o.flatten(1)))       # dino.py:142-144
    pred = v(img)
    h1.remove(); h2.remove()
    assert tuple(got["latents"].shape) == (2, 17, 32) and tuple(got["hidden"].shape) == (2, 32)
    assert _rel(got["hidden"], got["latents"][:, 0]) < 1e-6        # pool = cls: to_latent sees the cls row of the transformer output
    assert _rel(pred, plain) < 1e-5
    assert _rel(v.mlp_head(got["hidden"]), plain) < 1e-5


def _torch_block(t, x, r, dim, heads, dim_head):
    """The block of `t` in plain torch (vit.py:51-64, 18-25, 78-83), float64: output and input gradient for the cotangent r."""
    B, N = x.shape[:2]
    attn, ff = t.layers[0]
    xd = x.detach().double().requires_grad_(True)
    ln = lambda z, m: torch.nn.functional.layer_norm(z, (dim,), m.weight.double(), m.bias.double())
    h = ln(xd, attn.norm)
    q, k, vv = (h @ attn.to_qkv.weight.double().t()).chunk(3, dim=-1)
    sp = lambda z: z.view(B, N, heads, dim_head).transpose(1, 2)
    a = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) * dim_head ** -0.5, dim=-1) @ sp(vv)
    a = a.transpose(1, 2).reshape(B, N, heads * dim_head)
    if not isinstance(attn.to_out, torch.nn.Identity):
        a = a @ attn.to_out[0].weight.double().t() + attn.to_out[0].bias.double()
    x1 = a + xd
    f = ff.net
    h2 = torch.nn.functional.gelu(ln(x1, f[0]) @ f[1].weight.double().t() + f[1].bias.double())
    x2 = h2 @ f[4].weight.double().t() + f[4].bias.double() + x1
    ref = ln(x2, t.norm)
    (ref * r.double()).sum().backward()
    return ref, xd.grad


@gpu
def test_standalone_transformer_as_t2t_and_mae_use_it():
    """t2t.py:45 builds Transformer(dim = d, heads = 1, depth = 1, dim_head = d, mlp_dim = d) with d = 3 * 7 * 7 = 147 and
    147 * 9 = 1323: one head as wide as the model, an Identity output projection, and widths that are NOT multiples of 4 (they
    run op by op on the any-width kernels); mae.py:37 builds an ordinary decoder Transformer."""
    from vit_pytorch_amd.vit import Transformer
    torch.manual_seed(0)
    for dim, heads, dim_head in ((147, 1, 147), (1323, 1, 1323), (144, 1, 144), (64, 4, 16)):
        t = Transformer(dim=dim, depth=1, heads=heads, dim_head=dim_head, mlp_dim=dim).to("cuda")
        x = torch.randn(2, 49, dim, device="cuda", requires_grad=True)
        r = torch.randn(2, 49, dim, device="cuda")      # a generic cotangent (mean(y^2) of a LayerNorm output with gamma = 1, beta = 0
        y = t(x)                                        # has an analytically ZERO input gradient: pure cancellation, not a test)
        (y * r).sum().backward()
        ref, gref = _torch_block(t, x, r, dim, heads, dim_head)
        assert _rel(y, ref) < 1e-5 and _rel(x.grad, gref) < 1e-4, (dim, _rel(y, ref), _rel(x.grad, gref))
        for p_ in t.parameters():
            assert p_.grad is not None and torch.isfinite(p_.grad).all()


@gpu
def test_t2t_width_in_bfloat16():
    """The same odd-width block with 16-bit parameters: every op of the op-by-op path has an element-at-a-time form."""
    from vit_pytorch_amd.vit import Transformer
    torch.manual_seed(1)
    dim = 147
    t32 = Transformer(dim=dim, depth=1, heads=1, dim_head=dim, mlp_dim=dim).to("cuda")
    t = Transformer(dim=dim, depth=1, heads=1, dim_head=dim, mlp_dim=dim).to("cuda")
    t.load_state_dict(t32.state_dict())
    t = t.to(torch.bfloat16)
    x = torch.randn(2, 81, dim, device="cuda")
    r = torch.randn(2, 81, dim, device="cuda")
    xb = x.to(torch.bfloat16).requires_grad_(True)
    y = t(xb)
    (y.float() * r).sum().backward()
    ref, gref = _torch_block(t32, x, r, dim, 1, dim)
    assert _rel(y, ref) < 3e-2 and _rel(xb.grad, gref) < 6e-2, (_rel(y, ref), _rel(xb.grad, gref))


@needs_ref
def test_reference_t2t_vit_builds_on_the_drop_in_transformer():
    """t2t.py:5 imports vit_pytorch.vit.Transformer; with the alias its token-to-token layers (t2t.py:45) and its main
    transformer (t2t.py:57) are the drop-in's modules, including the 147- and 1323-wide ones."""
    import vit_pytorch_amd.vit as mine
    mod = _load_ref_module("t2t")
    v = mod.T2TViT(image_size=64, num_classes=10, dim=64, depth=2, heads=2, mlp_dim=128)
    ts = [m for m in v.modules() if isinstance(m, mine.Transformer)]
    assert len(ts) == 3
    assert sorted(t.norm.weight.shape[0] for t in ts) == [64, 147, 1323]
    for t in ts[:2]:
        assert isinstance(t.layers[0][0].to_out, torch.nn.Identity)      # heads == 1 and dim_head == dim (vit.py:45)
