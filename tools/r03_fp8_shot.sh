#!/bin/bash
# Round 3's last GPU call (8 GPU-minutes left): everything the fp8 work needs measured, most important first; every step
# writes its own log under gpurun_out/ so that a cut-off call still leaves the earlier results.
#   bash tools/r03_fp8_shot.sh
set -u
root=$PWD
export PYTHONPATH=$root
out=$root/gpurun_out
mkdir -p $out
t0=$(date +%s)
stamp() { echo "[shot] +$(( $(date +%s) - t0 )) s: $*"; }

stamp "fp8 tests (kernels, state machine on the device, training step vs the reference goldens)"
timeout 200 python -m pytest tests/test_fp8_backward_gpu.py tests/test_fp8_forward_gpu.py tests/test_fp8_gpu.py -q -m gpu -s -p no:cacheprovider > $out/shot_fp8_tests.log 2>&1
echo "rc=$?" >> $out/shot_fp8_tests.log
grep -E "passed|failed|error" $out/shot_fp8_tests.log | tail -3
grep -E "^FAILED|^ERROR" $out/shot_fp8_tests.log | head -40

stamp "config 5 (ViT-H/14 @ 336, batch 256), fp8, K = 32 forms"
VITK_FP8_K128=0 timeout 120 python bench.py --config vit_h14 --fp8 --steps 3 --warmup 3 --repeats 1 --no-cpu-baseline > $out/shot_h14_fp8_k32.log 2>&1
tail -1 $out/shot_h14_fp8_k32.log | cut -c1-600

stamp "fp8 kernel flavours, TF/s"
timeout 90 python tools/fp8_kernels_bench.py 64 > $out/shot_fp8_kernels.log 2>&1
tail -40 $out/shot_fp8_kernels.log | cut -c1-200

stamp "config 5, fp8, K = 128 form"
VITK_FP8_K128=1 timeout 120 python bench.py --config vit_h14 --fp8 --steps 3 --warmup 3 --repeats 1 --no-cpu-baseline > $out/shot_h14_fp8_k128.log 2>&1
tail -1 $out/shot_h14_fp8_k128.log | cut -c1-600

stamp "config 5, bf16 (same box, for the ratio)"
timeout 120 python bench.py --config vit_h14 --precision bf16 --steps 3 --warmup 2 --repeats 1 --no-cpu-baseline > $out/shot_h14_bf16.log 2>&1
tail -1 $out/shot_h14_bf16.log | cut -c1-400

stamp "the whole GPU suite"
timeout 300 python -m pytest tests -q -m gpu -x -p no:cacheprovider > $out/shot_gpu_tests.log 2>&1
tail -3 $out/shot_gpu_tests.log

stamp "headline bench (bf16, unchanged path)"
timeout 200 python bench.py --no-cpu-baseline > $out/shot_bench.log 2>&1
tail -1 $out/shot_bench.log | cut -c1-400
stamp done
