#!/bin/bash
export PYTHONPATH=$PWD
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gemm_persist_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r2c_test_persist.log
grep -v amdgpu.ids gpurun_out/r2c_test_persist.log
VITK_NTP_EPIS=31 timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" 2>&1 | tail -8 > gpurun_out/r2c_test_kernels.log
grep -v amdgpu.ids gpurun_out/r2c_test_kernels.log
timeout 300 python tools/nt_ab.py 5 > gpurun_out/r2c_ab.log 2>&1; grep -v amdgpu.ids gpurun_out/r2c_ab.log | cut -c1-160
VITK_NTP_NOEXACT=1 timeout 300 python tools/nt_ab.py 3 > gpurun_out/r2c_ab_noexact.log 2>&1; grep -v amdgpu.ids gpurun_out/r2c_ab_noexact.log | cut -c1-160
