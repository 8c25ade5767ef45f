#!/bin/bash
# round 4, GPU call 34: the LDS-DMA varlen attention kernels (csrc/attention_varlen.hip): tests, then old / new (geometry 1, 2) on one box
set -u
root=$PWD; export PYTHONPATH=$root; out=$root/gpurun_out/r04f; mkdir -p $out
timeout 600 python -m pytest tests/test_navit_gpu.py -x -q -k "varlen or pool_geometry" > $out/r04f_varlen_tests.log 2>&1; echo "tests rc=$?" >> $out/r04f_varlen_tests.log
tail -5 $out/r04f_varlen_tests.log
log=$out/r04f_vl_bench.log; : > $log
for rep in 1 2; do
VITK_LIB=$root/vit_pytorch_amd/libvitk_oldvl.so timeout 300 python tools/vl_bench.py 64 >> $log 2>>$out/err.log
VITK_ATTN_VL=2 timeout 300 python tools/vl_bench.py 64 >> $log 2>>$out/err.log
VITK_ATTN_VL=1 timeout 300 python tools/vl_bench.py 64 >> $log 2>>$out/err.log
done
cat $log; tail -5 $out/err.log
