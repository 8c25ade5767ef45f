#!/bin/bash
# round 4, GPU call 46: the tree rebuilt from scratch in a new container: whole GPU suite + the default bench line
set -u
root=$PWD; export PYTHONPATH=$root; out=$root/gpurun_out/r04j; mkdir -p $out
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $out/r04j_gpu_tests.log; cat $out/r04j_gpu_tests.log
timeout 400 python bench.py > $out/r04j_bench.json.log 2> $out/r04j_bench.err; tail -1 $out/r04j_bench.json.log | cut -c1-600
