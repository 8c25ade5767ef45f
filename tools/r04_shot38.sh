#!/bin/bash
# round 4, GPU call 38: varlen attention with the XCD-aware (block, head) order: tests, times at geometry 1 / 2, ablations
set -u
root=$PWD; export PYTHONPATH=$root; out=$root/gpurun_out/r04f; mkdir -p $out
export VITK_LIB=$root/vit_pytorch_amd/libvitk_e0.so
timeout 600 python -m pytest tests/test_navit_gpu.py -x -q -k "varlen or pool_geometry" 2>&1 | tail -3
log=$out/r04f_vl_xcd.log; : > $log
for g in 1 2; do
VITK_ATTN_VL=$g timeout 300 python tools/vl_bench.py 64 >> $log 2>>$out/err.log
done
for g in 1 2; do for d in 1 2; do
VL_BENCH_ONLY=h14 VITK_ATTN_VL=$g VITK_VL_DBG=$d timeout 120 python tools/vl_bench.py 64 >> $log 2>>$out/err.log
done; done
VITK_ATTN_VL=1 timeout 300 python tools/vl_bench.py 256 >> $log 2>>$out/err.log
cat $log
