#!/bin/bash
# round 4, GPU call 35: varlen attention -- transposing reads issued at the top of the step (e1) vs in front of their MFMAs (e0), geometry 1 / 2
set -u
root=$PWD; export PYTHONPATH=$root; out=$root/gpurun_out/r04f; mkdir -p $out
log=$out/r04f_vl_bench_early.log; : > $log
for rep in 1 2; do
for lib in e0 e1; do for g in 1 2; do
VITK_LIB=$root/vit_pytorch_amd/libvitk_$lib.so VITK_ATTN_VL=$g timeout 300 python tools/vl_bench.py 64 >> $log 2>>$out/err.log
done; done
done
cat $log
VITK_LIB=$root/vit_pytorch_amd/libvitk_e1.so timeout 600 python -m pytest tests/test_navit_gpu.py -x -q -k "varlen or pool_geometry" 2>&1 | tail -3
