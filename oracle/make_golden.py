"""Generate tests/golden/*.npz by EXECUTING THE REFERENCE in the build container.

TEST INFRASTRUCTURE ONLY.  Runs only where /root/reference exists (it does not
travel to the GPU box); the fixtures it writes are committed.

    python oracle/make_golden.py [case ...]

For every case in oracle/params.py::CASES it loads the unmodified reference
module by file path (vit.py / simple_vit.py import only torch + einops;
importing the package would pull torchvision via __init__.py:5 -> dino.py:9),
constructs the model, loads the deterministic state_dict, runs
forward + ``loss_fn`` + backward in fp32 on the CPU and stores

    logits            the model output
    grad::<key>       d loss / d parameter for every state_dict key
    loss              the scalar

Weights and inputs are NOT stored: they are regenerated from the seed by
oracle/params.py (numpy RNG, platform independent).
"""
from __future__ import annotations

import importlib.util
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True  # never write __pycache__ into /root/reference
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle.params import (CASES, NAVIT_BENCH_CASES, NAVIT_CASES, NAVIT_WIDE_CASES, navit_bench_sizes, VARIANT_CASES, WIDE_CASES, make_params_for, make_images, make_navit_images, make_navit_params, make_params,  # noqa: E402
                           sample_index)
from oracle.vit_oracle import loss_fn  # noqa: E402

REF = "/root/reference/vit_pytorch"
ONLY = set(sys.argv[1:])      # optional case names: regenerate just those (the others keep their committed bytes)


def load_ref(name: str):
    spec = importlib.util.spec_from_file_location(f"_ref_{name}", os.path.join(REF, f"{name}.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def run_reference(kind: str, cfg: dict, params, img, dtype=torch.float32):
    mod = load_ref("vit" if kind == "vit" else "simple_vit")
    cls = mod.ViT if kind == "vit" else mod.SimpleViT
    model = cls(**cfg)
    missing = model.load_state_dict(params, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    model.train()  # dropout p = 0 in every case, so train == eval numerically
    if dtype != torch.float32:
        model = model.to(dtype); img = img.to(dtype)
    out = model(img)
    loss = loss_fn(out)
    loss.backward()
    grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in model.named_parameters()}
    return out.detach(), loss.detach(), grads


def main():
    torch.manual_seed(0)
    torch.set_num_threads(1)  # fixed reduction order
    outdir = os.path.join(os.path.dirname(HERE), "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    for name, case in CASES.items():
        if ONLY and name not in ONLY:
            continue
        params = make_params(case["kind"], case["cfg"], case["seed"])
        img = make_images(case["cfg"], case["batch"], case["seed"] + 1000, case.get("image"))
        out, loss, grads = run_reference(case["kind"], case["cfg"], params, img)
        blob = {"logits": out.numpy(), "loss": loss.numpy()}
        for k, g in grads.items():
            blob["grad::" + k] = g.numpy()
        path = os.path.join(outdir, name + ".npz")
        np.savez(path, **blob)
        print(f"{name}: logits {tuple(out.shape)} loss {float(loss):.6f} "
              f"{len(grads)} grads -> {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def main_wide():
    """Compact goldens at the layer shapes of BASELINE configs 2 / 3 / 5 (oracle/params.py::WIDE_CASES): the reference in f32 and
    -- as the yardstick a 16-bit pipeline is held against -- the reference's OWN pure-bf16 run on the same inputs."""
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    outdir = os.path.join(os.path.dirname(HERE), "tests", "golden")
    for name, case in WIDE_CASES.items():
        if ONLY and name not in ONLY:
            continue
        params = make_params(case["kind"], case["cfg"], case["seed"])
        img = make_images(case["cfg"], case["batch"], case["seed"] + 1000)
        out, loss, grads = run_reference(case["kind"], case["cfg"], params, img)
        out16, _, grads16 = run_reference(case["kind"], case["cfg"], params, img, torch.bfloat16)
        blob = {"logits": out.numpy(), "loss": loss.numpy(), "bf16::logits": out16.float().numpy()}
        for k, g in grads.items():
            idx = sample_index(g.numel(), case.get("sample", 4096))
            blob["gnorm::" + k] = np.float64(g.double().norm().item())
            blob["gsample::" + k] = g.flatten().numpy()[idx]
            blob["bf16::gsample::" + k] = grads16[k].float().flatten().numpy()[idx]
        path = os.path.join(outdir, name + ".npz")
        np.savez_compressed(path, **blob)
        print(f"{name}: logits {tuple(out.shape)} loss {float(loss):.6f} {len(grads)} grads -> {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


AUTOCAST_CASES = ("vit_b16_width", "vit_b16_full", "vit_b16_full_b256")       # (the last: ~40 GiB of host memory, the headline run itself)


def main_autocast():
    """The reference under torch.autocast (what `accelerate`'s mixed precision runs, train_vit_decorr.py:74-78): float32 master parameters,
    Linear / matmul in bfloat16, LayerNorm / softmax / the residual stream in float32.  For the WIDE_CASES named in AUTOCAST_CASES the
    unmodified reference is executed inside torch.autocast("cpu", dtype=torch.bfloat16) on the same parameters and images as the float32
    golden; logits and the same gradient samples go to tests/golden/<case>__autocast.npz.  The drop-in's autocast route is gated against
    THESE numbers (tests/test_autocast_parity_gpu.py): at most 1.5x the reference-autocast's own distance from the float32 run + 1e-3."""
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    outdir = os.path.join(os.path.dirname(HERE), "tests", "golden")
    for name in AUTOCAST_CASES:
        if ONLY and name + "__autocast" not in ONLY:
            continue
        case = WIDE_CASES[name]
        params = make_params(case["kind"], case["cfg"], case["seed"])
        img = make_images(case["cfg"], case["batch"], case["seed"] + 1000)
        mod = load_ref("vit" if case["kind"] == "vit" else "simple_vit")
        model = (mod.ViT if case["kind"] == "vit" else mod.SimpleViT)(**case["cfg"])
        model.load_state_dict(params, strict=True)
        model.train()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            out = model(img)
        loss = loss_fn(out.float())
        loss.backward()
        blob = {"autocast::logits": out.detach().float().numpy(), "autocast::logits_dtype": np.frombuffer(str(out.dtype).encode(), dtype=np.uint8)}
        for k, p_ in model.named_parameters():
            g = p_.grad if p_.grad is not None else torch.zeros_like(p_)
            idx = sample_index(g.numel(), case.get("sample", 4096))
            blob["autocast::gsample::" + k] = g.detach().float().flatten().numpy()[idx]
        path = os.path.join(outdir, name + "__autocast.npz")
        np.savez_compressed(path, **blob)
        print(f"{name} under autocast(bf16): logits {tuple(out.shape)} {out.dtype} -> {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def main_variants():
    """The sibling variants (oracle/params.py::VARIANT_CASES): reference in eval mode (dropouts off), f32; the golden also records
    the reference's state_dict keys and shapes, which the drop-in's module must reproduce."""
    import json
    from collections import OrderedDict
    outdir = os.path.join(os.path.dirname(HERE), "tests", "golden")
    for name, case in VARIANT_CASES.items():
        if ONLY and name not in ONLY:
            continue
        mod = load_ref(case["module"])
        model = getattr(mod, case["cls"])(**case["cfg"])
        shapes = OrderedDict((k, tuple(v.shape)) for k, v in model.state_dict().items())
        params = make_params_for(shapes, case["seed"])
        model.load_state_dict(params, strict=True)
        model.eval()
        img = make_images(case["cfg"], case["batch"], case["seed"] + 1000)
        out = model(img)
        loss = loss_fn(out)
        loss.backward()
        blob = {"logits": out.detach().numpy(), "loss": loss.detach().numpy(),
                "state_dict_shapes": np.frombuffer(json.dumps([[k, list(v)] for k, v in shapes.items()]).encode(), dtype=np.uint8)}
        for k, p in model.named_parameters():
            blob["grad::" + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
        path = os.path.join(outdir, name + ".npz")
        np.savez_compressed(path, **blob)
        print(f"{name}: logits {tuple(out.shape)} loss {float(loss):.6f} -> {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def main_navit():
    outdir = os.path.join(os.path.dirname(HERE), "tests", "golden")
    mod = load_ref("na_vit")
    for name, case in NAVIT_CASES.items():
        if ONLY and name not in ONLY:
            continue
        params = make_navit_params(case["cfg"], case["seed"])
        images = make_navit_images(case["cfg"], case["sizes"], case["seed"] + 1000)
        model = mod.NaViT(**case["cfg"])
        model.load_state_dict(params, strict=True)
        model.eval()  # no token dropout / dropout in the parity case
        out = model(images)
        loss = loss_fn(out)
        loss.backward()
        blob = {"logits": out.detach().numpy(), "loss": loss.detach().numpy()}
        for k, p in model.named_parameters():
            blob["grad::" + k] = p.grad.numpy()
        path = os.path.join(outdir, name + ".npz")
        np.savez(path, **blob)
        print(f"{name}: logits {tuple(out.shape)} loss {float(loss):.6f} -> {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def main_navit_wide():
    """NaViT at BASELINE config 4's width, one 4,096-token pack (oracle/params.py::NAVIT_WIDE_CASES): compact golden, f32 and the
    reference's own bf16 run."""
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    outdir = os.path.join(os.path.dirname(HERE), "tests", "golden")
    mod = load_ref("na_vit")
    for name, case in NAVIT_WIDE_CASES.items():
        if ONLY and name not in ONLY:
            continue
        params = make_navit_params(case["cfg"], case["seed"])
        images = make_navit_images(case["cfg"], case["sizes"], case["seed"] + 1000)
        assert sum((h // 16) * (w // 16) for (h, w) in case["sizes"][0]) == case.get("tokens", 4096)
        res = {}
        for dtype in (torch.float32, torch.bfloat16):
            model = mod.NaViT(**case["cfg"])
            model.load_state_dict(params, strict=True)
            model = model.to(dtype).eval()
            out = model([[im.to(dtype) for im in pack] for pack in images])
            loss_fn(out.float()).backward()
            res[dtype] = (out.detach().float(), {k: p.grad.detach().float() for k, p in model.named_parameters()})
        out, grads = res[torch.float32]
        out16, grads16 = res[torch.bfloat16]
        blob = {"logits": out.numpy(), "bf16::logits": out16.numpy()}
        for k, g in grads.items():
            idx = sample_index(g.numel(), case["sample"])
            blob["gnorm::" + k] = np.float64(g.double().norm().item())
            blob["gsample::" + k] = g.flatten().numpy()[idx]
            blob["bf16::gsample::" + k] = grads16[k].flatten().numpy()[idx]
        path = os.path.join(outdir, name + ".npz")
        np.savez_compressed(path, **blob)
        print(f"{name}: logits {tuple(out.shape)} {len(grads)} grads -> {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def main_navit_bench():
    """NaViT on the BENCH workload's image draw (oracle/params.py::NAVIT_BENCH_CASES): a flat list of 65 images, grouped by the
    reference itself (group_images=True, group_max_seq_len=4096); compact golden, f32 and the reference's own bf16 run."""
    import gc
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    outdir = os.path.join(os.path.dirname(HERE), "tests", "golden")
    mod = load_ref("na_vit")
    for name, case in NAVIT_BENCH_CASES.items():
        if ONLY and name not in ONLY:
            continue
        sizes = navit_bench_sizes()
        params = make_navit_params(case["cfg"], case["seed"])
        images = make_navit_images(case["cfg"], [sizes], case["seed"] + 1000)[0]        # one flat list
        res = {}
        for dtype in (torch.float32, torch.bfloat16):
            model = mod.NaViT(**case["cfg"])
            model.load_state_dict(params, strict=True)
            model = model.to(dtype).eval()
            out = model([im.to(dtype) for im in images], group_images=True, group_max_seq_len=case["group_max_seq_len"])
            loss_fn(out.float()).backward()
            res[dtype] = (out.detach().float(), {k: p.grad.detach().float() for k, p in model.named_parameters()})
            del model, out
            gc.collect()
        out, grads = res[torch.float32]
        out16, grads16 = res[torch.bfloat16]
        blob = {"logits": out.numpy(), "bf16::logits": out16.numpy(), "sizes": np.asarray(sizes, dtype=np.int64)}
        for k, g in grads.items():
            idx = sample_index(g.numel(), case["sample"])
            blob["gnorm::" + k] = np.float64(g.double().norm().item())
            blob["gsample::" + k] = g.flatten().numpy()[idx]
            blob["bf16::gsample::" + k] = grads16[k].flatten().numpy()[idx]
        path = os.path.join(outdir, name + ".npz")
        np.savez_compressed(path, **blob)
        print(f"{name}: {len(sizes)} images, logits {tuple(out.shape)} {len(grads)} grads -> {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    if ONLY and all(n in NAVIT_BENCH_CASES for n in ONLY):
        main_navit_bench()
        sys.exit(0)
    if ONLY and all(n.endswith("__autocast") for n in ONLY):
        main_autocast()
        sys.exit(0)
    main()
    main_navit()
    main_wide()
    main_variants()
    main_navit_wide()
    main_navit_bench()
    main_autocast()
