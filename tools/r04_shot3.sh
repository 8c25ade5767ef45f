#!/bin/bash
# round 4, GPU call 3: the w128 TN kernel in the library -- its tests, the TN A/B at the step's shapes, the step
set -u
root=$PWD; export PYTHONPATH=$root; out=$root/gpurun_out/r04_shot3; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_headline_extents_gpu.py tests/test_abi.py -x -q -m gpu -k "tn or TN or dw or headline or linear" > $out/tests.log 2>&1
tail -5 $out/tests.log
timeout 300 python bench.py --no-cpu-baseline > $out/bench.log 2>$out/bench.err
tail -1 $out/bench.log | cut -c1-300
VITK_FWD_STREAM=16 timeout 300 python bench.py --no-cpu-baseline > $out/bench_s16.log 2>$out/bench_s16.err
tail -1 $out/bench_s16.log | cut -c1-300
VITK_TN_W128=0 timeout 300 python bench.py --no-cpu-baseline > $out/bench_w128off.log 2>$out/bench_w128off.err
tail -1 $out/bench_w128off.log | cut -c1-300
