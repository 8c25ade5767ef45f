#!/bin/bash
export PYTHONPATH=$PWD
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm_tn" 2>&1 | tail -8 | grep -v amdgpu
timeout 300 python tools/tn_ab.py 5 2>&1 | grep -v amdgpu | tee gpurun_out/r2f_tn_ab.log
