#!/bin/bash
export PYTHONPATH=$PWD
mkdir -p gpurun_out
for dbg in 0 1 2 3; do
VITK_NTP_DBG=$dbg timeout 300 python tools/nt_ab.py 3 > gpurun_out/r2b_ab_dbg$dbg.log 2>&1; grep -v amdgpu.ids gpurun_out/r2b_ab_dbg$dbg.log | cut -c1-150
done
