#!/bin/bash
export PYTHONPATH=$PWD
VITK_TN_XREG=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm_tn" 2>&1 | tail -4 | grep -v amdgpu
echo "== XREG=1"; VITK_TN_XREG=1 timeout 300 python tools/tn_ab.py 4 2>&1 | grep -v amdgpu | cut -c1-130
echo "== XREG=0"; timeout 300 python tools/tn_ab.py 2 2>&1 | grep -v amdgpu | cut -c1-130
echo "== XREG=1 no-MFMA"; VITK_TN_XREG=1 VITK_TN_DBG=2 timeout 300 python tools/tn_ab.py 2 2>&1 | grep -v amdgpu | cut -c1-130
