#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on BASELINE.json's config, measured on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
            --master-port P bench.py --gpus N --steps K --warmup W)

Workload ("step"): one forward + loss + backward of ViT-B/16 224x224, bf16, batch 256 PER GPU
(BASELINE.json configs[1]) on synthetic on-device images and labels, random-init
weights; when N > 1 the step includes the data-parallel gradient all-reduce (RCCL) overlapped with the
patch-embedding backward.  Optimizer excluded (the north star names fwd+bwd).  Weak scaling:
per-GPU batch fixed, `value` = images/s summed over all GPUs.

Timed region: W warm-up steps, then barrier + torch.cuda.synchronize(), exactly K steps,
synchronize + barrier; the time is the MAX over ranks.  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline      the step's large GEMMs are timed by CLASS on the launches of real training steps (every C-ABI call bracketed by HIP
                events on its stream, 3 extra steps after the timed region, weight-gradient side stream off so that a launch's duration
                is the kernel's, not the contention's).  `classes` lists every class with its kernel time per step, its share of the
                measured step, achieved TFLOP/s and its heaviest instance; the top-level achieved / frac / avg_launch_ms are the
                heaviest instance of the HEAVIEST class by time (round 4: the weight-gradient GEMM or the plain NT GEMMs, not FF1).
                peak = 2516.6 TFLOP/s dense bf16 MFMA.  `traffic` is STATIC (profiles/rNN_pmc_traffic.json, see `traffic_source`).
  weight_cache_rebuild_ms
                the timed steady state never changes the weights, so the K-blocked / transposed weight copies are built once;
                a real training step (optimizer after every backward) rebuilds them every step.  This is the extra time per
                step of a 10-step window that calls `invalidate_weight_caches()` before every step (what every step of a real
                run pays; round 6: all copies of the stack in one table launch, vitk_pack_w_nt_many), measured after the timed
                region against a steady window: `ms_per_step + weight_cache_rebuild_ms` is the fwd+bwd time inside a training loop.
  model         whole-step algorithmic TFLOP/s (105.383 GF/img for ViT-B/16) and its fraction of peak.
  box           a ~60 ms calibration of THIS box before the timed region, with plain torch ops (not the product): a bf16 torch.mm
                (hipBLASLt) of 8192^3 on random data in TFLOP/s, a 1 GiB fill and a 1 GiB copy in TB/s.  The pool's boxes differ by +-5 % on the
                same tree; these three numbers let a driver line from a slow box be read.  `value` is NOT normalised by them.
  fp16          the same model and kernels with IEEE-half parameters (libvitk_f16.so): images/s measured here (N = 1 only), and the
                parity numbers of both 16-bit types against the reference's float32 run on the headline configuration itself
                (static: profiles/r06_headline_parity.json, written by the GPU parity test of the batch-256 golden).  bfloat16 -- the
                dtype BASELINE.json quotes the metric in -- lands at the reference-bf16's own 9e-3; float16 meets the north star's 1e-3.
  cpu_baseline  the CPU oracle (oracle/vit_oracle.py, kind "port") timed on this host's cores on a bounded
                sample of the same workload (same model, f32, batch 32, best of a thread sweep), rank 0 at N = 1 only.

`--config vit_h14` is BASELINE config 5's line and runs in fp8 by default (`--precision bf16` for the 16-bit number): `dtype: "fp8"`
with `dtype_detail` saying which GEMMs run in which format (all twelve GEMMs of a layer on fp8 operands; attention and LayerNorm in
bf16) and `roof` naming the peak the
timed fp8 kernel is priced against: 5,033.2 TFLOP/s for the K = 128 form (the default), 2,516.6 for the non-scaled K = 32 fp8
forms (VITK_FP8_K128=0).  The default line (no flags) is the bf16 headline and is unchanged by any of this.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2516.6  # 256 CU x 4 SIMD x 1024 FLOP/clk x 2.4 GHz (MI355X_MICROARCH.md: ~2.5 PF dense)

CONFIGS = {
    # name: (ctor kwargs, per-GPU batch)
    "vit_b16": (dict(image_size=224, patch_size=16, num_classes=1000, dim=768, depth=12, heads=12, mlp_dim=3072), 256),
    "vit_l16": (dict(image_size=224, patch_size=16, num_classes=1000, dim=1024, depth=24, heads=16, mlp_dim=4096), 128),
    # config 5's architecture and per-GPU batch (2048 / 8); runs in fp8 by default (vit_pytorch_amd.fp8.enable_fp8: all twelve GEMMs of a
    # layer on fp8 operands), `--precision bf16` for the 16-bit number beside it (that one fits through engine._recompute_policy)
    "vit_h14": (dict(image_size=336, patch_size=14, num_classes=1000, dim=1280, depth=32, heads=16, dim_head=80, mlp_dim=5120), 256),
}


def fwd_gflop_per_image(cfg) -> float:
    """fwd = 2 Np P D + depth (2 N D 3I + 4 h N^2 dh + 2 N I D + 4 N D F) + 2 D C; fwd+bwd = 3x."""
    ph = cfg["patch_size"]
    Np = (cfg["image_size"] // ph) ** 2
    P = 3 * ph * ph
    D, depth, h, F, C = cfg["dim"], cfg["depth"], cfg["heads"], cfg["mlp_dim"], cfg["num_classes"]
    dh = cfg.get("dim_head", 64)
    I = h * dh
    N = Np + 1
    f = 2 * Np * P * D + depth * (2 * N * D * 3 * I + 4 * h * N * N * dh + 2 * N * I * D + 4 * N * D * F) + 2 * D * C
    return f / 1e9


def time_gemm_classes(step_fn, nsteps: int = 3):
    """Per-launch durations of every large GEMM of the step, by kernel class and shape, measured on the launches of the real training
    step: every call of the C-ABI entry point is bracketed by HIP events on the stream it is enqueued on.  Done in `nsteps` extra
    steps after the timed region, with the weight-gradient side stream OFF (beside a concurrent GEMM a launch's duration measures the
    contention, not the kernel).  Returns {(class, N, K): (mean ms, flops per launch, launches)}; classes:
      tn        weight gradient dW = dY^T X (gemm_tn_w128_kernel + the slab fold)
      ff1       NT GEMM with the fused bias + GELU epilogue       dff1   NT GEMM with the GELU' + bias-gradient column sums epilogue
      nt_resid  NT GEMM + bias + residual (out-projection, FF2)    nt     NT GEMM, plain / bias epilogue (QKV, the three dX GEMMs)"""
    from vit_pytorch_amd import _lib as L, kernels as K
    orig = {n: getattr(K, n) for n in ("gemm_nt_bf16", "gemm_nt_bf16_gelu_bwd_colsum", "gemm_nt_bf16_mul_aux_colsum", "gemm_nt_bf16_mul_aux8_colsum", "gemm_nt_fp8_v2", "gemm_tn_bf16",
                                       "gemm_tn_bf16_pair")}
    taps = {}

    def bracket(key, flops, fn, a, kw):
        st = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        r = fn(*a, **kw)
        e1.record(st)
        taps.setdefault(key, []).append((e0, e1, flops))
        return r

    nt_class = {L.EPI_BIAS_GELU: "ff1", L.EPI_BIAS_GELU_DG: "ff1", L.EPI_BIAS_GELU_DG8: "ff1", L.EPI_MUL_AUX8: "dff1", L.EPI_GELU_BWD: "dff1", L.EPI_MUL_AUX: "dff1", L.EPI_RESID: "nt_resid", L.EPI_RESID16: "nt_resid"}

    def tapped_nt(*a, **kw):
        epi = a[9] if len(a) > 9 else kw.get("epilogue", L.EPI_NONE)
        return bracket((nt_class.get(epi, "nt"), a[7], a[8]), 2.0 * a[6] * a[7] * a[8], orig["gemm_nt_bf16"], a, kw)

    def tapped_bwd(*a, **kw):
        return bracket(("dff1", a[7], a[8]), 2.0 * a[6] * a[7] * a[8], orig["gemm_nt_bf16_gelu_bwd_colsum"], a, kw)

    def tapped_mul(*a, **kw):
        return bracket(("dff1", a[7], a[8]), 2.0 * a[6] * a[7] * a[8], orig["gemm_nt_bf16_mul_aux_colsum"], a, kw)

    def tapped_f8(*a, **kw):        # --fp8: the forward / dX GEMMs go through the fp8 entry point (a[9] = epilogue)
        return bracket((nt_class.get(a[9], "nt"), a[7], a[8]), 2.0 * a[6] * a[7] * a[8], orig["gemm_nt_fp8_v2"], a, kw)

    def tapped_tn(*a, **kw):
        return bracket(("tn", a[7], a[8]), 2.0 * a[6] * a[7] * a[8], orig["gemm_tn_bf16"], a, kw)

    K.gemm_nt_bf16, K.gemm_nt_bf16_gelu_bwd_colsum, K.gemm_nt_fp8_v2, K.gemm_tn_bf16 = tapped_nt, tapped_bwd, tapped_f8, tapped_tn
    K.gemm_nt_bf16_mul_aux_colsum = tapped_mul

    def tapped_mul8(*a, **kw):
        return bracket(("dff1", a[7], a[8]), 2.0 * a[6] * a[7] * a[8], orig["gemm_nt_bf16_mul_aux8_colsum"], a, kw)
    K.gemm_nt_bf16_mul_aux8_colsum = tapped_mul8

    def tapped_pair(*a, **kw):          # (dY0, ldy0, X0, ldx0, dW0, dY1, ldy1, X1, ldx1, dW1, M, ws, splits): two weight gradients, one launch
        (n0, k0), (n1, k1) = a[4].shape, a[9].shape
        return bracket(("tn", f"{n0}+{n1}", f"{k0}|{k1}"), 2.0 * a[10] * (n0 * k0 + n1 * k1), orig["gemm_tn_bf16_pair"], a, kw)
    K.gemm_tn_bf16_pair = tapped_pair
    from vit_pytorch_amd import engine as E
    E._Fork.serialize = True          # serialized launches (the fp8 path's side stream off): engine._Fork reads it per backward
    try:
        for _ in range(nsteps):
            step_fn()
        torch.cuda.synchronize()
    finally:
        for n, f in orig.items():
            setattr(K, n, f)
        E._Fork.serialize = False
    return {key: (sum(e0.elapsed_time(e1) for e0, e1, _ in tl) / len(tl), tl[0][2], len(tl) // nsteps) for key, tl in taps.items()}


TRAFFIC_FILES = ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04h_pmc_traffic.json", "r04d_pmc_traffic.json", "r04c_pmc_traffic.json", "r04b_pmc_traffic.json", "r04_pmc_traffic.json", "r03_final_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json")
# (class, N, K) of a ViT-B/16 GEMM -> the label tools/kprof.py / tools/pmc_traffic_json.py give its launch group
TRAFFIC_LABELS = {("tn", 3072, 768): "dW ff1", ("tn", 2304, 768): "dW qkv", ("ff1", 3072, 768): "FF1 bias+GELU", ("dff1", 3072, 768): "dFF1 GELU'",
                  ("nt", 2304, 768): "QKV", ("nt_resid", 768, 3072): "FF2 +", ("nt_resid", 768, 768): "out-proj +", ("nt", 768, 3072): "dX of FF1"}


def pmc_traffic_bytes(key):
    """HBM bytes per launch of a GEMM instance from the newest committed PMC collection (profiles/rNN_pmc_traffic.json: rocprofv3
    --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md).
    Counters cannot be collected from inside this process: the value is STATIC (taken from that file, not from this run);
    (None, None, None) if no file has the instance.  Returns (traffic bytes, algorithmic bytes, source)."""
    label = TRAFFIC_LABELS.get(key)
    if label is None:
        return None, None, None
    for name in TRAFFIC_FILES:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                ks = json.load(f)["kernels"]
            v = next(v for k, v in ks.items() if label in k)
            return v["traffic_bytes"], v.get("algorithmic_bytes"), f"profiles/{name} (static: collected in a separate rocprofv3 --pmc run)"
        except Exception:
            continue
    return None, None, None


def cpu_baseline(cfg, budget_s: float = 30.0):
    """The oracle (a port of the reference algorithm, kind "port") on the host cores at batch 32, swept over
    intra-op thread counts because oversubscribing the host is slower than using a fraction of it; the best setting is
    reported with its thread count.  Bounded to ~budget_s in total."""
    from oracle import vit_oracle as O
    from oracle.params import make_images, make_params
    ncpu = os.cpu_count() or 1
    params = make_params("vit", cfg, 0)
    B = 32
    img = make_images(cfg, B, 1)
    p = {k: v.clone().requires_grad_(v.numel() > 0) for k, v in params.items()}

    def step():
        for v in p.values():
            v.grad = None
        out = O.vit_fwd(img, p, patch_size=cfg["patch_size"], depth=cfg["depth"], heads=cfg["heads"],
                        dim_head=cfg.get("dim_head", 64), pool="cls", num_classes=cfg["num_classes"])
        O.loss_fn(out).backward()

    default_threads = torch.get_num_threads()
    sweep = [t for t in (16, 32, 8, 64) if t <= ncpu] or [min(default_threads, ncpu)]      # most promising first: the budget may cut the tail
    results = {}
    t_begin = time.perf_counter()
    try:
        for t in sweep:
            if time.perf_counter() - t_begin > budget_s:
                break
            torch.set_num_threads(t)
            step()                                   # warm-up at this thread count
            t0 = time.perf_counter()
            n = 0
            while n < 3 and (n == 0 or time.perf_counter() - t_begin < budget_s):
                step()
                n += 1
            results[t] = B * n / (time.perf_counter() - t0)
    finally:
        torch.set_num_threads(default_threads)
    best = max(results, key=results.get)
    return {"value": round(results[best], 3), "unit": "images/s", "cores": best, "kind": "port",
            "sample": f"oracle/vit_oracle.py ViT-B/16 fwd+bwd f32, batch {B}, thread sweep " +
                      ", ".join(f"{t}: {v:.2f} img/s" for t, v in results.items()) + f" on a host with {ncpu} cpus "
                      f"({time.perf_counter() - t_begin:.0f} s in total)"}


def bench_navit(args, dev):
    """BASELINE config 4 (`--config navit`): a packed variable-resolution batch -- images a*16 x b*16 px with a, b ~ U{4..40}
    (64-640 px) until ~32k tokens, grouped into packs of <= 4096 tokens, NaViT dim 1024 / depth 24 / heads 16 / mlp 4096, bf16,
    forward + cross-entropy + backward.  One JSON line; the metric is tokens/s (images/s alongside), single GPU only."""
    import numpy as np
    from vit_pytorch_amd.na_vit import NaViT
    rng = np.random.default_rng(0)
    sizes, tok = [], 0
    while tok < 32768:
        a, b = rng.integers(4, 41, 2)
        sizes.append((int(a) * 16, int(b) * 16)); tok += int(a) * int(b)
    depth, D, F, H, I = 24, 1024, 4096, 16, 1024
    torch.manual_seed(0)
    m = NaViT(image_size=1024, patch_size=16, num_classes=1000, dim=D, depth=depth, heads=H, mlp_dim=F).to(dev, dtype=torch.bfloat16)
    imgs = [torch.randn(3, h, w, device=dev).to(torch.bfloat16) for h, w in sizes]
    labels = torch.randint(0, 1000, (len(imgs),), device=dev)

    def step():
        m.zero_grad(set_to_none=True)
        out = m(imgs, group_images=True, group_max_seq_len=4096)
        torch.nn.functional.cross_entropy(out.float(), labels).backward()

    for _ in range(args.warmup):
        step()
    dts = []
    for _ in range(max(1, args.repeats)):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize(dev)
        dts.append(time.perf_counter() - t0)
    dt = sorted(dts)[len(dts) // 2] / args.steps
    lens = [(h // 16) * (w // 16) for h, w in sizes]
    gemm = depth * tok * (2 * D * 3 * I + 2 * I * D + 4 * D * F)
    attn = depth * sum(4 * H * n * n * 64 for n in lens)
    tf = 3 * (gemm + attn) / dt / 1e12
    print(json.dumps({
        "metric": "tokens/sec (fwd+bwd) NaViT packed variable-resolution batch, bf16", "value": round(tok / dt, 1), "unit": "tokens/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 3),
        "ms_per_step_all": [round(d / args.steps * 1e3, 3) for d in dts], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic (randn images of random sizes, randint labels, random-init weights)",
        "config": {"workload": f"navit dim {D} depth {depth} heads {H} mlp {F}: {len(imgs)} images of 64-640 px, {tok} tokens, packs of <= 4096 tokens, "
                               "cross-entropy loss, no optimizer", "images": len(imgs), "tokens": tok, "max_tokens_per_image": max(lens), "parallelism": "dp1"},
        "images_per_s": round(len(imgs) / dt, 2),
        "model": {"tflops_per_gpu": round(tf, 2), "frac_of_mfma_peak": round(tf / PEAK_BF16_TFLOPS, 4),
                  "attention_share_of_flops": round(attn / (gemm + attn), 4)},
    }), flush=True)


def box_calibration(dev):
    """Plain torch ops, outside the timed region: how fast is THIS box?  (bf16 matmul on random data, a fill, a copy.)"""
    def timed(fn, iters):
        fn(); torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) / iters * 1e-3
    try:
        n = 8192
        a = torch.randn(n, n, device=dev).to(torch.bfloat16); b = torch.randn(n, n, device=dev).to(torch.bfloat16)
        c = torch.empty(n, n, device=dev, dtype=torch.bfloat16)
        t_mm = timed(lambda: torch.mm(a, b, out=c), 20)
        x = torch.empty(1 << 30, device=dev, dtype=torch.uint8); y = torch.empty_like(x)
        t_fill = timed(lambda: x.fill_(1), 10)
        t_copy = timed(lambda: y.copy_(x), 10)
        del a, b, c, x, y
        return {"torch_mm_bf16_8192_tflops": round(2.0 * n ** 3 / t_mm / 1e12, 1), "fill_1gib_tb_s": round((1 << 30) / t_fill / 1e12, 2),
                "copy_1gib_tb_s": round(2.0 * (1 << 30) / t_copy / 1e12, 2),
                "note": "plain torch ops (hipBLASLt matmul on random data, fill_, copy_: read + write bytes), timed before the benchmark; not part of the product, not used to normalise `value`"}
    except Exception as e:       # a calibration failure must not cost the benchmark line
        return {"error": repr(e)}


def fp16_leg(cfg, batch, dev, steps: int = 10):
    """The same step with IEEE-half parameters (libvitk_f16.so: the same sources, the 16-bit type switched); N = 1 only."""
    from vit_pytorch_amd import ViT
    torch.manual_seed(0)
    m = ViT(**cfg).to(dev, dtype=torch.float16)
    g = torch.Generator(device=dev); g.manual_seed(1)
    img = torch.randn(batch, 3, cfg["image_size"], cfg["image_size"], device=dev, generator=g).to(torch.float16)
    labels = torch.randint(0, cfg["num_classes"], (batch,), device=dev, generator=g)

    def step():
        m.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(m(img).float(), labels)
        (loss * 1024.0).backward()       # a constant loss scale, as any float16 training uses
        return loss
    for _ in range(3):
        step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    out = {"images_per_s": round(batch / dt, 2), "ms_per_step": round(dt * 1e3, 3), "loss_finite": bool(torch.isfinite(loss).item())}
    try:
        with open(os.path.join(ROOT, "profiles", "r06_headline_parity.json")) as f:
            out["parity_vs_reference_f32"] = json.load(f)
    except Exception:
        out["parity_vs_reference_f32"] = None
    del m
    return out


def rccl_version():
    try:
        return ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        return None


def spawn_command(n: int, argv, port: int):
    """The launch line `python bench.py --gpus N ...` turns itself into when no launcher set WORLD_SIZE: exactly the driver's own
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__), *argv]


def self_spawn(n: int) -> int:
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n:
        print(json.dumps({"error": f"--gpus {n} but torch.cuda.device_count() = {have}", "n_gpus": n, "device_count": have}), flush=True)
        return 2
    with socket.socket() as s:               # a free rendezvous port on the loopback
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(spawn_command(n, sys.argv[1:], port), env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="vit_b16", choices=list(CONFIGS) + ["navit"])
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch override (invalidates the headline number)")
    ap.add_argument("--no-cpu-baseline", action="store_true",
                    help="skip the side measurements -- cpu_baseline, the fp16 leg and the box calibration -- so that a profiled run contains the timed workload only")
    ap.add_argument("--precision", choices=["bf16", "fp8"], default=None,
                    help="GEMM operand precision; default: the one BASELINE.json quotes the config in (vit_b16 / vit_l16: bf16, vit_h14 = config 5: fp8)")
    ap.add_argument("--fp8", action="store_true",
                    help="BASELINE config 5's precision (vit_pytorch_amd.fp8.enable_fp8): e4m3 operands for the four forward GEMMs of a "
                         "layer, e5m2 gradients for its four dX and four dW GEMMs; attention and LayerNorm stay bf16.  Same as "
                         "--precision fp8.  VITK_FP8_K128=0 selects the K = 32 MFMA forms (default: K = 128).  Not the headline metric (that one is bf16).")
    ap.add_argument("--repeats", type=int, default=3,
                    help="time the K-step region this many times and report the MEDIAN window (every repeat is exactly K steps "
                         "between barrier + synchronize; all of them are listed in ms_per_step_all)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_spawn(args.gpus))      # plain `python bench.py --gpus N`: re-exec under torch.distributed.run, one rank per GPU
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if args.config == "navit":
        if world > 1:
            raise SystemExit("--config navit is a single-GPU benchmark")
        bench_navit(args, dev)
        return

    from vit_pytorch_amd import ViT
    from vit_pytorch_amd.parallel import DataParallel

    cfg, batch = CONFIGS[args.config]
    if args.precision is None:
        args.precision = "fp8" if (args.fp8 or args.config == "vit_h14") else "bf16"
    args.fp8 = args.precision == "fp8"
    if args.batch:
        batch = args.batch
    torch.manual_seed(0)  # identical init on every rank (DataParallel also broadcasts rank 0's)
    model = ViT(**cfg).to(dev, dtype=torch.bfloat16)
    if args.fp8:
        from vit_pytorch_amd.fp8 import enable_fp8
        enable_fp8(model)
        args.warmup = max(args.warmup, 3)       # steps 1-2 decide the delayed scales (16-bit recording passes); timed steps are steady state
    dp = DataParallel(model)
    g = torch.Generator(device=dev)
    g.manual_seed(1 + rank)
    img = torch.randn(batch, 3, cfg["image_size"], cfg["image_size"], device=dev, generator=g).to(torch.bfloat16)
    labels = torch.randint(0, cfg["num_classes"], (batch,), device=dev, generator=g)

    def step():
        logits = dp(img)
        loss = torch.nn.functional.cross_entropy(logits.float(), labels)
        dp.backward(loss)  # zeroes .grad, backward, overlapped all-reduce when world > 1
        return loss

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    box = box_calibration(dev) if (rank == 0 and not args.no_cpu_baseline) else None
    for _ in range(args.warmup):
        step()
    # Shared GPU boxes show occasional multi-x slow phases (clock / power state; one measured run: 100.9, 39.6, 39.6 ms);
    # one such phase inside a single 20-step window would misreport the kernel work, so the window is repeated and the
    # median window reported (all windows are in ms_per_step_all).
    dts, rank_dts = [], []
    for _ in range(max(1, args.repeats)):
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            every = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(every, t)
            rank_dts.append([float(e.item()) for e in every])
            dt = max(rank_dts[-1])                   # the step is as slow as its slowest rank
        else:
            rank_dts.append([dt])
        dts.append(dt)
    mid = sorted(range(len(dts)), key=dts.__getitem__)[len(dts) // 2]
    dt = dts[mid]
    assert torch.isfinite(loss).item(), "loss is not finite"
    comm_stats = None
    if world > 1:       # one more step with the collectives timed (HIP events on the communication stream) -- after the timed region
        dp.sink.enable_timing(True)
        step()
        comm_stats = dp.sink.timing_stats()
        dp.sink.enable_timing(False)
    taps = time_gemm_classes(step)      # every rank: the steps contain the collectives

    # one step with every weight-derived cache invalidated (= what each step of a real training loop pays) vs a steady step
    from vit_pytorch_amd import invalidate_weight_caches

    # (windows of 10 steps, the CPU running ahead of the GPU as in a real loop: a single synchronised step would add the host-side
    #  allocation of the copies to the GPU's time)
    def window_ms(invalidate: bool, n: int = 10) -> float:
        sync()
        t0 = time.perf_counter()
        for _ in range(n):
            if invalidate:
                invalidate_weight_caches()
            step()
        sync()
        return (time.perf_counter() - t0) * 1e3 / n
    window_ms(True, 2)
    rebuild = [window_ms(True) - window_ms(False) for _ in range(3)]
    weight_cache_rebuild_ms = sorted(rebuild)[1]

    if rank == 0:
        ms = dt / args.steps * 1e3
        total_imgs = batch * world * args.steps
        value = total_imgs / dt
        gf = 3.0 * fwd_gflop_per_image(cfg)
        N = (cfg["image_size"] // cfg["patch_size"]) ** 2 + 1
        k128 = bool(args.fp8 and model.transformer._fp8.k128)
        peak = PEAK_BF16_TFLOPS * (2.0 if k128 else 1.0)
        # GEMM classes by their share of the step (kernel time per step / measured step time); the roofline kernel is the heaviest
        # INSTANCE (one shape) of the heaviest class
        CLASS_NAMES = {"tn": "gemm_tn_w128_kernel + tn_reduce_kernel (weight gradients dW = dY^T X: four waves, 128 x 128 wave tiles)",
                       "ff1": "gemm_ntw_kernel<EPI_BIAS_GELU_DG8> (four-wave persistent NT GEMM, FF1: bias + GELU, stores the gelu' factor for the backward as 8-bit codes) "
                              "[+ gemm_ntp_kernel on the rows the row split leaves]",
                       "dff1": "gemm_ntw_kernel<EPI_MUL_AUX8> (dFF1: x the stored gelu' factor + bias-gradient column sums) [+ gemm_ntp_kernel on the remaining rows]",
                       "nt_resid": "gemm_ntw_kernel<EPI_RESID16 / EPI_RESID> + gemm_ntp_kernel on the remaining rows (out-projection, FF2: + bias + residual)",
                       "nt": "gemm_ntw_kernel<EPI_NONE / EPI_BIAS> + gemm_ntp_kernel on the remaining rows (QKV and the three dX GEMMs)"}
        if args.fp8:
            CLASS_NAMES = {k: v.replace("gemm_ntw_kernel", "gemm_nt256pp_kernel (fp8 operands)").replace("gemm_ntp_kernel", "gemm_nt256pp_kernel (fp8 operands)").replace("gemm_tn_w128_kernel", "gemm_tn_fp8") for k, v in CLASS_NAMES.items()}
        classes = {}
        for (cls, n_, k_), (kms_, kfl_, kn_) in taps.items():
            c = classes.setdefault(cls, {"ms": 0.0, "flops": 0.0, "launches": 0, "inst": []})
            c["ms"] += kms_ * kn_; c["flops"] += kfl_ * kn_; c["launches"] += kn_
            c["inst"].append(((cls, n_, k_), kms_, kfl_, kn_))

        balance = round(peak * 1e12 / 6.3e12, 1)

        def inst_entry(key, kms_, kfl_, kn_):
            tr, alg, src = (None, None, None) if args.fp8 else pmc_traffic_bytes(key)
            ach_ = kfl_ / (kms_ * 1e-3) / 1e12
            e = {"shape": f"N={key[1]} K={key[2]} (M = {batch * N} token rows)", "achieved": round(ach_, 2), "frac": round(ach_ / peak, 4),
                 "avg_launch_ms": round(kms_, 4), "launches_per_step": kn_, "traffic": tr, "algorithmic_bytes": alg, "traffic_source": src}
            if alg and kfl_:       # which roof the bytes say: FLOP per algorithmic byte against the machine balance (peak / 6.3 TB/s)
                e["flop_per_byte"] = round(kfl_ / alg, 1)
                e["hbm_frac"] = round(alg / (kms_ * 1e-3) / 6.3e12, 4)      # every instance is priced against BOTH roofs (6.3 TB/s achievable)
            return e

        ranked = sorted(classes.items(), key=lambda kv: -kv[1]["ms"])
        dom_cls, dom = ranked[0]
        dkey, dms_, dfl_, dn_ = max(dom["inst"], key=lambda t: t[1] * t[3])
        dent = inst_entry(dkey, dms_, dfl_, dn_)
        dom_ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12          # the CLASS: all its launches of the step (not its best instance)
        dom_avg_ms = dom["ms"] / dom["launches"]
        others = []
        for cls, c in ranked:
            ach_c = c["flops"] / (c["ms"] * 1e-3) / 1e12
            top = max(c["inst"], key=lambda t: t[1] * t[3])
            others.append({"class": cls, "kernel": CLASS_NAMES.get(cls, cls), "ms_per_step": round(c["ms"], 3), "share_of_step": round(c["ms"] / ms, 4),
                           "launches_per_step": c["launches"], "achieved": round(ach_c, 2), "frac": round(ach_c / peak, 4),
                           "heaviest_instance": inst_entry(*top)})
        isz = cfg["image_size"]
        prec = "fp8" if args.fp8 else "bf16"
        # the roof a number is priced against: bf16 MFMA dense for the headline; with --fp8 the fp8 forms the GEMMs actually issue --
        # K = 32 (v_mfma_f32_16x16x32_fp8_fp8 / _fp8_bf8) run at the bf16 rate, K = 128 (v_mfma_f32_16x16x128_f8f6f4) at twice that
        roof_name = ("fp8 MFMA dense, K = 128 f8f6f4 form: 2 x 2516.6 = 5033.2 TFLOP/s" if k128 else
                     "fp8 MFMA dense, non-scaled K = 32 forms: 2516.6 TFLOP/s (= the bf16 rate)" if args.fp8 else "bf16 MFMA dense: 2516.6 TFLOP/s")
        line = {
            "metric": "images/sec (fwd+bwd) ViT-B/16 224^2 bf16" if (args.config == "vit_b16" and not args.fp8) else f"images/sec (fwd+bwd) {args.config} {prec}",
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "ms_per_step_all": [round(d / args.steps * 1e3, 3) for d in dts],
            "weight_cache_rebuild_ms": round(weight_cache_rebuild_ms, 3),
            "train_loop_ms_per_step": round(ms + weight_cache_rebuild_ms, 3),      # fwd+bwd inside a loop whose optimizer changes the weights every step
            "train_loop_images_per_s": round(batch * world / ((ms + weight_cache_rebuild_ms) * 1e-3), 2),
            "box": box,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": prec, "data": "synthetic (randn images, randint labels, random-init weights)",
            **({"dtype_detail": "e4m3 x e4m3 forward GEMMs (QKV, out-projection, FF1, FF2), e5m2 gradients x e4m3 weights^T for the four dX GEMMs, "
                                "e5m2 gradients^T x e4m3 activations for the four dW GEMMs, per-tensor delayed scaling, f32 accumulation; attention, "
                                "LayerNorm and the residual streams in bf16 / f32" + ("; K = 128 MFMA" if k128 else "; K = 32 MFMA forms"),
                "roof": roof_name} if args.fp8 else {}),
            "memory": {"peak_allocated_gib": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1),
                       "peak_reserved_gib": round(torch.cuda.max_memory_reserved(dev) / 2 ** 30, 1),
                       "allocator_retries": int(torch.cuda.memory_stats(dev).get("num_alloc_retries", 0))},
            "config": {"workload": f"{args.config} fwd+bwd, batch {batch}/GPU, {isz}x{isz}, cross-entropy loss, no optimizer"
                                   + (", flat-buffer RCCL all-reduce overlapped with patch-embed backward" if world > 1 else ""),
                       "global_batch": batch * world, "per_gpu_batch": batch, "seq_len": N, "parallelism": f"dp{world}"},
            "per_gpu_images_per_s": round(value / world, 2),
            "per_rank_images_per_s": [round(batch * args.steps / d, 2) for d in rank_dts[mid]],       # each rank's own clock, the reported window
            "device_count": torch.cuda.device_count(), "rccl_version": rccl_version(),
            **({"comm": {**comm_stats, "process_group_world": dist.get_world_size(),
                         "note": "one extra step after the timed region: bytes and event-timed duration of each gradient all-reduce on the "
                                 "communication stream, and exposed_ms = how long the main stream waited for it at the end of backward"}}
               if comm_stats else {}),
            "model": {"gflop_per_image_fwd_bwd": round(gf, 3), "tflops_per_gpu": round(value / world * gf / 1e3, 2),
                      "frac_of_mfma_peak": round(value / world * gf / 1e3 / PEAK_BF16_TFLOPS, 4),
                      **({"frac_note": "whole-step fraction is of the bf16 peak 2516.6 TF/s (the number comparable with the bf16 line); of the fp8 roof named in `roof` it is "
                                       + str(round(value / world * gf / 1e3 / peak, 4))} if args.fp8 else {})},
            "roofline": {"bound": "mfma", "kernel": CLASS_NAMES.get(dom_cls, dom_cls) + f": heaviest class of the step ({round(dom['ms'] / ms * 100, 1)} % of it); "
                                                      f"achieved / frac / avg_launch_ms are the CLASS's (algorithmic flops of all its launches / their summed durations), "
                                                      f"heaviest_instance is {dent['shape']}",
                         "achieved": round(dom_ach, 2), "peak": peak, "unit": "TFLOP/s", "roof": roof_name, "frac": round(dom_ach / peak, 4),
                         "avg_launch_ms": round(dom_avg_ms, 4), "launches_timed": dom["launches"] * 3,
                         "heaviest_instance": dent,
                         "worst_class": (lambda wc: {"class": wc["class"], "frac": wc["frac"], "achieved": wc["achieved"], "ms_per_step": wc["ms_per_step"],
                                                     **({"hbm_frac": wc["heaviest_instance"]["hbm_frac"]} if "hbm_frac" in wc["heaviest_instance"] else {})})(
                                            min(others, key=lambda o: o["frac"])),
                         "timed_in": "3 extra steps after the timed region, weight-gradient side stream off (serialized launches); a class's share_of_step = its "
                                     "serialized kernel time per step / the measured step",
                         "traffic": dent["traffic"], "algorithmic_bytes": dent["algorithmic_bytes"], "traffic_source": dent["traffic_source"],
                         "machine_balance_flop_per_byte": balance,
                         "bound_note": "MFMA roof; an instance whose flop_per_byte is below machine_balance_flop_per_byte (peak / 6.3 TB/s) is balanced or HBM-side by its bytes "
                                       "(FF1 at K = 768 with a 16-bit and an 8-bit output: 435 FLOP/B against 399; with two 16-bit outputs it was 339)",
                         "classes": others},
        }
        if world == 1 and args.config == "vit_b16" and not args.fp8 and not args.batch and not args.no_cpu_baseline:
            del dp, model
            torch.cuda.empty_cache()
            line["fp16"] = fp16_leg(cfg, batch, dev)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
