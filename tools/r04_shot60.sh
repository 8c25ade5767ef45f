#!/bin/bash
# round 4, final GPU call of the round: whole GPU suite on the end-of-round tree + the default bench line + serialized kernel stats
set -u
root=$PWD; export PYTHONPATH=$root; out=$root/gpurun_out/r04k; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -15 > $out/r04k_gpu_tests.log; grep -E "passed|failed" $out/r04k_gpu_tests.log
timeout 300 python bench.py > $out/r04k_bench.json.log 2> $out/r04k_bench.err; tail -1 $out/r04k_bench.json.log | cut -c1-300
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke > $out/r04k_smoke.log; tail -2 $out/r04k_smoke.log
