#!/bin/bash
# round 4, GPU call 40: varlen attention, the three workgroup orders (VITK_VL_ORDER 0 / 1 / 2) on the dense config-5 shape (batch 256) and the NaViT mix
set -u
root=$PWD; export PYTHONPATH=$root; out=$root/gpurun_out/r04f; mkdir -p $out
export VITK_LIB=$root/vit_pytorch_amd/libvitk_e0.so
log=$out/r04f_vl_orders.log; : > $log
for o in 0 1 2; do for g in 1 2; do
echo "== order $o geometry $g" >> $log
VITK_VL_ORDER=$o VITK_ATTN_VL=$g timeout 300 python tools/vl_bench.py 256 >> $log 2>>$out/err.log
done; done
VITK_VL_ORDER=2 timeout 600 python -m pytest tests/test_navit_gpu.py -x -q -k "varlen or pool_geometry" 2>&1 | tail -2 >> $log
cat $log
