#!/bin/bash
export PYTHONPATH=$PWD
VITK_NTP_PIPE=1 timeout 600 python -m pytest tests/test_gemm_persist_gpu.py -x -q 2>&1 | tail -5 | grep -v amdgpu
echo "== PIPE=1"; VITK_NTP_PIPE=1 timeout 300 python tools/nt_ab.py 4 2>&1 | grep -v amdgpu | cut -c1-150
echo "== PIPE=1 no epilogue"; VITK_NTP_PIPE=1 VITK_NTP_DBG=1 timeout 300 python tools/nt_ab.py 2 2>&1 | grep -v amdgpu | cut -c1-150
echo "== PIPE=0"; timeout 300 python tools/nt_ab.py 2 2>&1 | grep -v amdgpu | cut -c1-150
