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
    monkeypatch.delenv("VITK_WEIGHT_CACHE", raising=False)
    losses, final, ps = _train(opt_name, fp8)
    monkeypatch.setenv("VITK_WEIGHT_CACHE", "0")
    losses0, final0, ps0 = _train(opt_name, fp8)
    assert losses == losses0, (losses, losses0)
    assert torch.equal(final, final0)
    assert all(torch.equal(a, b) for a, b in zip(ps, ps0))
    assert losses[-1] < losses[0]                     # and it trains


# ---- usage patterns of a training script around the model ---------------------------------------------------------------------------
from oracle import vit_oracle as O  # noqa: E402
from vit_pytorch_amd import SimpleViT  # noqa: E402
from vit_pytorch_amd.parallel import DataParallel  # noqa: E402


def _rel(a, b):
    a = a.detach().double().flatten().cpu(); b = b.detach().double().flatten().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 2e-2)])
def test_gradient_accumulation_and_model_called_twice(dtype, tol):
    """Two micro-batches accumulated into .grad (backward twice, no zero_grad), and the model called twice inside ONE graph (siamese /
    DINO-style, dino.py:283-290), both equal the gradient of the summed loss computed call by call -- plain model and under
    DataParallel (whose flat buffer holds one slot per parameter)."""
    params = make_params("vit", CFG, 11)
    xa = make_images(CFG, 8, 3001).to(DEV, dtype=dtype); xb = make_images(CFG, 5, 3002).to(DEV, dtype=dtype)

    def fresh():
        m = ViT(**CFG); m.load_state_dict(params)
        return m.to(DEV, dtype=dtype)

    def grads(m):
        return torch.cat([p.grad.float().flatten() for p in m.parameters() if p.numel()])

    la = lambda o: o.float().square().mean()
    lb = lambda o: (o.float() - 1).square().mean()
    m = fresh(); la(m(xa)).backward(); ga = grads(m)
    m = fresh(); lb(m(xb)).backward(); gb = grads(m)
    want = ga + gb
    m = fresh(); la(m(xa)).backward(); lb(m(xb)).backward()
    assert _rel(grads(m), want) <= tol                                   # accumulation over micro-batches
    m = fresh(); (la(m(xa)) + lb(m(xb))).backward()
    assert _rel(grads(m), want) <= tol                                   # two calls, one graph
    m = fresh(); dp = DataParallel(m, broadcast=False)
    dp.backward(la(dp(xa)) + lb(dp(xb)))
    assert _rel(grads(m), want) <= tol                                   # ... with the gradients in the flat buffer


@pytest.mark.parametrize("kind", ["vit", "simple_vit"])
@pytest.mark.parametrize("cfg_name", ["small_patch_generic", "b16_gather"])
def test_input_image_gradient_matches_the_oracle(kind, cfg_name):
    """img.requires_grad_(): the reference differentiates through Rearrange + LayerNorm(patch_dim) (vit.py:100-101); so does the drop-in
    (vitk_unpatchify behind the LayerNorm backward) -- through the generic patchify path and through the gather-in-the-load path
    (16 x 16 x 3 patches, 16-bit), against torch autograd over the oracle's forward (f32: 1e-3; bf16: 3e-2)."""
    cfg = (dict(image_size=32, patch_size=4, num_classes=10, dim=64, depth=1, heads=2, dim_head=32, mlp_dim=128, channels=4) if cfg_name == "small_patch_generic"
           else dict(image_size=64, patch_size=16, num_classes=10, dim=128, depth=1, heads=2, dim_head=64, mlp_dim=256))
    params = make_params(kind, cfg, 23)
    img = make_images(cfg, 3, 1023)
    # reference gradient: autograd through the oracle's forward functions (float64)
    p64 = {k: v.double() for k, v in params.items()}
    x64 = img.double().requires_grad_(True)
    fwd = O.vit_fwd if kind == "vit" else O.simple_vit_fwd
    kw = dict(patch_size=cfg["patch_size"], depth=cfg["depth"], heads=cfg["heads"], dim_head=cfg["dim_head"])
    if kind == "vit":
        kw["pool"] = "cls"
    O.loss_fn(fwd(x64, p64, **kw)).backward()
    for dtype, tol in ((torch.float32, 1e-3), (torch.bfloat16, 3e-2)):
        m = (ViT if kind == "vit" else SimpleViT)(**cfg); m.load_state_dict(params); m = m.to(DEV, dtype=dtype)
        x = img.to(DEV, dtype=dtype).requires_grad_(True)
        O.loss_fn(m(x)).backward()
        assert x.grad is not None and x.grad.shape == x.shape
        assert _rel(x.grad, x64.grad) <= tol, (kind, cfg_name, dtype, _rel(x.grad, x64.grad))
        # and through the op-by-op route (a hook on the embedding's LayerNorm takes the stage off the fused Function)
        h = m.to_patch_embedding[1].register_forward_hook(lambda *a: None)
        x2 = img.to(DEV, dtype=dtype).requires_grad_(True)
        O.loss_fn(m(x2)).backward()
        h.remove()
        assert _rel(x2.grad, x64.grad) <= tol, (kind, cfg_name, dtype, "op by op")


@pytest.mark.parametrize("kind", ["vit", "simple_vit"])
def test_autocast_around_a_float32_model_runs_the_16bit_engine(kind, monkeypatch):
    """`with torch.autocast("cuda", dtype=torch.bfloat16): model(x)` on float32 master weights -- mixed-precision training as accelerate
    sets it up around the reference (train_vit_decorr.py:74).  The fused engine runs on 16-bit copies of the parameters
    (functional.autocast_aware).  With VITK_AUTOCAST_STREAM=16 (the residual streams in the parameter dtype, round 5's behaviour) the
    logits are bit-identical to those of the same weights in a bfloat16 model; they come back in
    bfloat16 like the reference's autocast Linear gives them, the gradients arrive in float32 on the master parameters and equal the
    bfloat16 model's (rounded to bf16 there), and a forward hook on the model fires once.  The DEFAULT (round 6: float32 residual streams,
    like the reference under autocast) is held against the reference's own autocast run in tests/test_autocast_parity_gpu.py."""
    monkeypatch.setenv("VITK_AUTOCAST_STREAM", "16")
    cfg = dict(CFG) if kind == "vit" else {k: v for k, v in CFG.items()}
    params = make_params(kind, cfg, 31)
    cls = ViT if kind == "vit" else SimpleViT
    img = make_images(cfg, 8, 3100)
    m32 = cls(**cfg); m32.load_state_dict(params); m32 = m32.to(DEV)
    m16 = cls(**cfg); m16.load_state_dict(params); m16 = m16.to(DEV, dtype=torch.bfloat16)
    fired = []
    h = m32.register_forward_hook(lambda mod, i, o: fired.append(o.dtype))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = m32(img.to(DEV))
        loss = out.float().square().mean()
    loss.backward()
    h.remove()
    ref = m16(img.to(DEV, dtype=torch.bfloat16))
    ref.float().square().mean().backward()
    assert fired == [torch.bfloat16] and out.dtype == torch.bfloat16
    assert torch.equal(out, ref)
    for (n, p), (_, q) in zip(m32.named_parameters(), m16.named_parameters()):
        if p.numel():
            assert p.grad is not None and p.grad.dtype == torch.float32, n
            assert torch.equal(p.grad.to(torch.bfloat16), q.grad), n
    # outside autocast the float32 model is the float32 (validation-accuracy) model it was
    out32 = m32(img.to(DEV))
    assert out32.dtype == torch.float32 and _rel(out32, ref) < 3e-2 and not torch.equal(out32.to(torch.bfloat16), ref)


def test_autocast_fp16_with_grad_scaler_trains():
    params = make_params("vit", CFG, 31)
    m = ViT(**CFG); m.load_state_dict(params); m = m.to(DEV)
    opt = torch.optim.AdamW(m.parameters(), lr=3e-4, fused=True)
    scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
    losses = []
    for it in range(4):
        x = make_images(CFG, 8, 2000 + it).to(DEV)
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            out = m(x)
            loss = out.float().square().mean()
        assert out.dtype == torch.float16
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        losses.append(loss.item())
    assert all(torch.isfinite(p).all() for p in m.parameters()) and losses[-1] < losses[0], losses


def test_autocast_navit():
    from oracle.params import make_navit_images, make_navit_params
    from vit_pytorch_amd.na_vit import NaViT
    cfg = dict(image_size=64, patch_size=8, num_classes=7, dim=64, depth=2, heads=2, mlp_dim=96)
    sizes = [[(32, 48), (16, 16), (64, 24)], [(40, 40), (8, 56)]]
    params = make_navit_params(cfg, 5)
    imgs = make_navit_images(cfg, sizes, 1005)
    m32 = NaViT(**cfg); m32.load_state_dict(params); m32 = m32.to(DEV).eval()
    m16 = NaViT(**cfg); m16.load_state_dict(params); m16 = m16.to(DEV, dtype=torch.bfloat16).eval()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = m32([[im.to(DEV) for im in g] for g in imgs])
    out.float().square().mean().backward()
    ref = m16([[im.to(DEV, dtype=torch.bfloat16) for im in g] for g in imgs])
    assert out.dtype == torch.bfloat16 and torch.equal(out, ref)
    assert all(p.grad is not None and p.grad.dtype == torch.float32 for p in m32.parameters() if p.numel())


def test_ema_teacher_updated_through_data_is_seen_every_step(monkeypatch):
    """The DINO / BYOL loop (dino.py:95-99,303): the teacher is a deepcopy of the student whose weights are moved by `.data` writes after
    every optimizer step -- invisible to the parameters' version counters.  The optimizer step in between bumps the weights epoch, so the
    teacher's derived weight copies are rebuilt before its next forward: same logits as the cache-free run, step by step."""
    import copy

    def run():
        params = make_params("vit", CFG, 11)
        student = ViT(**CFG); student.load_state_dict(params); student = student.to(DEV, dtype=torch.bfloat16)
        teacher = copy.deepcopy(student).requires_grad_(False)
        opt = torch.optim.SGD(student.parameters(), lr=0.05)
        outs = []
        for it in range(3):
            x = make_images(CFG, 8, 2000 + it).to(DEV, dtype=torch.bfloat16)
            with torch.no_grad():
                t = teacher(x)
            outs.append(t.clone())
            opt.zero_grad(set_to_none=True)
            (student(x).float() - t.float()).square().mean().add(student(x).float().square().mean()).backward()
            opt.step()
            for ps, pt in zip(student.parameters(), teacher.parameters()):
                pt.data.lerp_(ps.data, 0.5)
        return outs

    monkeypatch.delenv("VITK_WEIGHT_CACHE", raising=False)
    a = run()
    monkeypatch.setenv("VITK_WEIGHT_CACHE", "0")
    b = run()
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert not torch.equal(a[1], a[2])


def test_torch_compile_runs_the_drop_in_eagerly_on_the_gpu():
    """torch.compile(model) must not crash on the ctypes-backed fused stages (functional.eager_modules marks every module of the package
    `torch.compiler.disable`): same logits as the plain call, gradients arrive.  backend="eager": Dynamo's tracing is what is at stake."""
    params = make_params("vit", CFG, 11)
    m = ViT(**CFG); m.load_state_dict(params); m = m.to(DEV, dtype=torch.bfloat16)
    x = make_images(CFG, 8, 3300).to(DEV, dtype=torch.bfloat16)
    ref = m(x)
    cm = torch.compile(m, backend="eager")
    out = cm(x)
    out.float().square().mean().backward()
    assert torch.equal(out, ref) and all(p.grad is not None for p in m.parameters() if p.numel())


def test_autocast_around_the_standalone_transformer_block(monkeypatch):
    """vit.Transformer used on its own (T2T-ViT builds its layers from it, t2t.py:45,57) with float32 parameters inside an autocast
    region: the 16-bit engine on 16-bit parameter copies, like the whole model (bit-identical to the bfloat16 block with the streams in the
    parameter dtype, VITK_AUTOCAST_STREAM=16; the default float32 streams stay within the bf16 rounding of it)."""
    from vit_pytorch_amd.vit import Transformer
    torch.manual_seed(3)
    t32 = Transformer(dim=256, depth=2, heads=4, dim_head=64, mlp_dim=512).to(DEV)
    t16 = Transformer(dim=256, depth=2, heads=4, dim_head=64, mlp_dim=512).to(DEV, dtype=torch.bfloat16)
    t16.load_state_dict({k: v.to(torch.bfloat16) for k, v in t32.state_dict().items()})
    x = torch.randn(8, 197, 256, device=DEV)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out_f32_streams = t32(x)
    monkeypatch.setenv("VITK_AUTOCAST_STREAM", "16")
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = t32(x)
    out.float().square().mean().backward()
    ref = t16(x.to(torch.bfloat16))
    assert out.dtype == torch.bfloat16 and torch.equal(out, ref)
    assert out_f32_streams.dtype == torch.bfloat16 and _rel(out_f32_streams, ref) < 2e-2
    assert all(p.grad is not None and p.grad.dtype == torch.float32 for p in t32.parameters())
