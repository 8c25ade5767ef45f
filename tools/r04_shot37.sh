#!/bin/bash
# round 4, GPU call 37: ablations of the varlen attention kernels at the config-5 shape (VITK_VL_DBG: 1 = no DMA, 2 = no arithmetic, 4 = no barrier)
set -u
root=$PWD; export PYTHONPATH=$root; out=$root/gpurun_out/r04f; mkdir -p $out
export VITK_LIB=$root/vit_pytorch_amd/libvitk_e0.so VL_BENCH_ONLY=h14
log=$out/r04f_vl_ablation.log; : > $log
for g in 1 2; do for d in 0 1 2 4 5 6; do
VITK_ATTN_VL=$g VITK_VL_DBG=$d timeout 120 python tools/vl_bench.py 64 >> $log 2>>$out/err.log
done; done
cat $log
