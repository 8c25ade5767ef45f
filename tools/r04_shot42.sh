#!/bin/bash
# round 4, GPU call 42: the whole GPU suite on the tree with the rebuilt variable-length attention kernels + the default bench line
set -u
root=$PWD; export PYTHONPATH=$root; out=$root/gpurun_out/r04g; mkdir -p $out
timeout 1200 python -m pytest tests -q -m gpu > $out/r04g_gpu_tests.log 2>&1; echo "rc=$?" >> $out/r04g_gpu_tests.log; tail -4 $out/r04g_gpu_tests.log
timeout 400 python bench.py --no-cpu-baseline > $out/r04g_bench.json.log 2> $out/bench.err; tail -1 $out/r04g_bench.json.log | cut -c1-300
