#!/bin/bash
# round 4, GPU call 39: varlen attention, groups of 16 blocks x all heads per XCD range: tests, times at geometry 1 / 2, batch 64 and 256
set -u
root=$PWD; export PYTHONPATH=$root; out=$root/gpurun_out/r04f; mkdir -p $out
export VITK_LIB=$root/vit_pytorch_amd/libvitk_e0.so
timeout 600 python -m pytest tests/test_navit_gpu.py -x -q -k "varlen or pool_geometry" 2>&1 | tail -3
log=$out/r04f_vl_xcd_groups.log; : > $log
for g in 1 2; do
VITK_ATTN_VL=$g timeout 300 python tools/vl_bench.py 64 >> $log 2>>$out/err.log
done
for g in 1 2; do
VL_BENCH_ONLY=h14 VITK_ATTN_VL=$g timeout 300 python tools/vl_bench.py 256 >> $log 2>>$out/err.log
done
VL_BENCH_ONLY=h14 VITK_LIB=$root/vit_pytorch_amd/libvitk_oldvl.so timeout 300 python tools/vl_bench.py 256 >> $log 2>>$out/err.log
cat $log
