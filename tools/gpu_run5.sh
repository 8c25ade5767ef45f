#!/bin/bash
export PYTHONPATH=$PWD
mkdir -p gpurun_out
python tools/diag_transformer.py 2>&1 | grep -v amdgpu | cut -c1-200
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r2e_tests.log; grep -v amdgpu.ids gpurun_out/r2e_tests.log
