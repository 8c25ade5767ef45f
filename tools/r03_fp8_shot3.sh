#!/bin/bash
# Round 3, third fp8 GPU call (the round's last ~2 GPU-minutes): GELU' epilogue emitting the e5m2 copy, FF1 without its unused 16-bit output.
set -u
root=$PWD
export PYTHONPATH=$root
out=$root/gpurun_out
mkdir -p $out
t0=$(date +%s)
stamp() { echo "[shot3] +$(( $(date +%s) - t0 )) s: $*"; }
stamp "fp8 tests"
timeout 60 python -m pytest tests/test_fp8_backward_gpu.py tests/test_fp8_forward_gpu.py -q -m gpu -s -p no:cacheprovider > $out/shot3_fp8_tests.log 2>&1
grep -E "passed|failed|error" $out/shot3_fp8_tests.log | tail -3
grep -E "^FAILED|^ERROR" $out/shot3_fp8_tests.log | head -20
stamp "config 5 (fp8 default)"
timeout 60 python bench.py --config vit_h14 --steps 3 --warmup 3 --repeats 1 --no-cpu-baseline > $out/shot3_h14_fp8.log 2>&1
tail -1 $out/shot3_h14_fp8.log | cut -c1-330
stamp "ViT-L/16 (config 3's model, batch 128) fp8 and bf16"
timeout 40 python bench.py --config vit_l16 --fp8 --steps 5 --warmup 3 --repeats 1 --no-cpu-baseline > $out/shot3_l16_fp8.log 2>&1
tail -1 $out/shot3_l16_fp8.log | cut -c1-330
timeout 40 python bench.py --config vit_l16 --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline > $out/shot3_l16_bf16.log 2>&1
tail -1 $out/shot3_l16_bf16.log | cut -c1-330
stamp done
