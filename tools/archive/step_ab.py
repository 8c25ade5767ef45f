"""Interleaved A/B of the whole training step (forward + cross-entropy + backward) under two settings of switches that are read per
call (Python-side: VITK_FWD_STREAM, VITK_GRAD_STREAM, VITK_DW_STREAM, VITK_FP8_LEAN, VITK_RECOMPUTE, ...), in ONE process on ONE box
-- boxes of the pool differ by a few percent, so only such pairs say anything.

    python tools/step_ab.py "VITK_FWD_STREAM=f32" "VITK_FWD_STREAM=16"                  # round 4, first thing: the 16-bit forward stream
    python tools/step_ab.py --config vit_h14 --fp8 "VITK_FP8_LEAN=1" "VITK_FP8_LEAN=0"
    options: --config vit_b16|vit_l16|vit_h14, --batch N, --fp8, --rounds R (default 5), --steps S per round and setting (default 5)

Prints the median step time of each setting, every round's pair, and the logits' distance between the two settings."""
import argparse
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import CONFIGS  # noqa: E402
from vit_pytorch_amd import ViT  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("a"); ap.add_argument("b")
    ap.add_argument("--config", default="vit_b16", choices=list(CONFIGS))
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--fp8", action="store_true")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    settings = [dict(kv.split("=", 1) for kv in s.split(",") if kv) for s in (args.a, args.b)]
    cfg, batch = CONFIGS[args.config]
    batch = args.batch or batch
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = ViT(**cfg).to(dev, dtype=torch.bfloat16)
    if args.fp8:
        from vit_pytorch_amd.fp8 import enable_fp8
        enable_fp8(model)
    img = torch.randn(batch, 3, cfg["image_size"], cfg["image_size"], device=dev).to(torch.bfloat16)
    labels = torch.randint(0, cfg["num_classes"], (batch,), device=dev)

    def apply(st):
        for k, v in st.items():
            os.environ[k] = v

    def step():
        model.zero_grad(set_to_none=True)
        logits = model(img)
        torch.nn.functional.cross_entropy(logits.float(), labels).backward()
        return logits

    outs = []
    for st in settings:                       # warm-up (and, with --fp8, the recording steps) under both settings
        apply(st)
        for _ in range(4):
            o = step()
        outs.append(o.detach().float().clone())
    torch.cuda.synchronize()
    times = ([], [])
    for r in range(args.rounds):
        for i, st in enumerate(settings):
            apply(st)
            step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            times[i].append((time.perf_counter() - t0) / args.steps * 1e3)
        print(f"round {r}: A {times[0][-1]:8.3f} ms   B {times[1][-1]:8.3f} ms", flush=True)
    a, b = statistics.median(times[0]), statistics.median(times[1])
    d = ((outs[0] - outs[1]).norm() / outs[0].norm()).item()
    print(f"A [{args.a}] median {a:.3f} ms;  B [{args.b}] median {b:.3f} ms;  B / A = {b / a:.4f};  logits differ by {d:.2e} (relative L2)")


if __name__ == "__main__":
    main()
