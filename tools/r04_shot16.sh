#!/bin/bash
# round 4, GPU call 16: carried first fragments (new libvitk.so) vs the committed kernel (libvitk_base.so), interleaved
set -u
root=$PWD; export PYTHONPATH=$root; out=$root/gpurun_out/r04_shot16; mkdir -p $out
for rep in 1 2 3; do
  VITK_LIB=$root/vit_pytorch_amd/libvitk_base.so timeout 120 python tools/nt_shapes.py 3 >> $out/times.log 2>&1
  VITK_LIB=$root/vit_pytorch_amd/libvitk.so timeout 120 python tools/nt_shapes.py 3 >> $out/times.log 2>&1
done
grep -v amdgpu.ids $out/times.log
