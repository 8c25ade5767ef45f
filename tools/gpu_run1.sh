#!/bin/bash
# round-2 GPU call 1: persistent NT kernel correctness + A/B timing
export PYTHONPATH=$PWD
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gemm_persist_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r2a_test_persist.log
cat gpurun_out/r2a_test_persist.log
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" 2>&1 | tail -8 > gpurun_out/r2a_test_kernels.log
cat gpurun_out/r2a_test_kernels.log
timeout 300 python tools/nt_ab.py 5 > gpurun_out/r2a_ab.log 2>&1; cat gpurun_out/r2a_ab.log
VITK_NTP_NOEXACT=1 timeout 300 python tools/nt_ab.py 3 > gpurun_out/r2a_ab_noexact.log 2>&1; cat gpurun_out/r2a_ab_noexact.log
VITK_NTP_TAIL=-1 timeout 300 python tools/nt_ab.py 3 > gpurun_out/r2a_ab_notail.log 2>&1; cat gpurun_out/r2a_ab_notail.log
