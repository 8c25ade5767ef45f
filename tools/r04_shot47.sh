#!/bin/bash
# round 4, GPU call 47: seeded random configurations (tests/test_fuzz_gpu.py) against the oracle
set -u
root=$PWD; export PYTHONPATH=$root; out=$root/gpurun_out/r04j; mkdir -p $out
timeout 900 python -m pytest tests/test_fuzz_gpu.py -q -s 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" > $out/r04j_fuzz.log; grep -E "passed|failed|FAILED|Error" $out/r04j_fuzz.log | head -60
