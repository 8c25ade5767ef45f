#!/bin/bash
# round 4, GPU call 43: varlen forward with the step in phases (joint decision on the reference maximum) vs the previous commit
set -u
root=$PWD; export PYTHONPATH=$root; out=$root/gpurun_out/r04g; mkdir -p $out
log=$out/r04g_vl_fwd_phases.log; : > $log
timeout 600 python -m pytest tests/test_navit_gpu.py -x -q -k "varlen or pool_geometry" 2>&1 | tail -2 >> $log
for rep in 1 2; do for lib in libvitk_prev.so libvitk.so; do
VITK_LIB=$root/vit_pytorch_amd/$lib timeout 300 python tools/vl_bench.py 256 >> $log 2>>$out/err.log
done; done
cat $log
