#!/bin/bash
# Round 3, second (last) fp8 GPU call: lean saving (the forward's e4m3 copies feed the weight-gradient GEMMs), K = 128 default,
# rocprofv3 kernel stats of config 5 in fp8.  Most important first; each step has its own log under gpurun_out/.
set -u
root=$PWD
export PYTHONPATH=$root
out=$root/gpurun_out
mkdir -p $out
t0=$(date +%s)
stamp() { echo "[shot2] +$(( $(date +%s) - t0 )) s: $*"; }

stamp "fp8 tests"
timeout 120 python -m pytest tests/test_fp8_backward_gpu.py tests/test_fp8_forward_gpu.py -q -m gpu -s -p no:cacheprovider > $out/shot2_fp8_tests.log 2>&1
grep -E "passed|failed|error" $out/shot2_fp8_tests.log | tail -3
grep -E "^FAILED|^ERROR" $out/shot2_fp8_tests.log | head -20

stamp "config 5 (default precision = fp8, K = 128, lean saving)"
timeout 100 python bench.py --config vit_h14 --steps 3 --warmup 3 --repeats 1 --no-cpu-baseline > $out/shot2_h14_fp8.log 2>&1
tail -1 $out/shot2_h14_fp8.log | cut -c1-330

stamp "rocprofv3 kernel stats of the same command"
( cd /tmp && export TMPDIR=/tmp && timeout 150 rocprofv3 --kernel-trace --stats -d $out/shot2_stats -o run --output-format csv -- python $root/bench.py --config vit_h14 --steps 2 --warmup 3 --repeats 1 --no-cpu-baseline > $out/shot2_stats.log 2>&1 )
cp $out/shot2_stats/*/run_kernel_stats.csv $out/shot2_h14_fp8_kernel_stats.csv 2>/dev/null || cp $out/shot2_stats/run_kernel_stats.csv $out/shot2_h14_fp8_kernel_stats.csv 2>/dev/null
rm -rf $out/shot2_stats
head -12 $out/shot2_h14_fp8_kernel_stats.csv | cut -c1-160

stamp "16-bit model parity (the saved-activation tuple changed shape)"
timeout 100 python -m pytest tests/test_parity_gpu.py tests/test_memory_gpu.py -q -m gpu -x -p no:cacheprovider -k "production_widths or recompute or memory or no_grad" > $out/shot2_parity.log 2>&1
tail -2 $out/shot2_parity.log

stamp "config 5 fp8 with the activations re-quantised in the backward (VITK_FP8_LEAN=0), same box"
VITK_FP8_LEAN=0 timeout 100 python bench.py --config vit_h14 --steps 3 --warmup 3 --repeats 1 --no-cpu-baseline > $out/shot2_h14_fp8_requant.log 2>&1
tail -1 $out/shot2_h14_fp8_requant.log | cut -c1-330

stamp "config 2's model (ViT-B/16, batch 256) in fp8 -- not the headline metric"
timeout 100 python bench.py --fp8 --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline > $out/shot2_b16_fp8.log 2>&1
tail -1 $out/shot2_b16_fp8.log | cut -c1-330
stamp done
