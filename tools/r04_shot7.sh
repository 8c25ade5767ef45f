#!/bin/bash
# round 4, GPU call 7: do trickled global stores cost the LDS-DMA stream anything when the storing waves never wait on vmcnt?
set -u
root=$PWD; out=$root/gpurun_out/r04_shot7; mkdir -p $out
export TN_PROBE_VARS=0,4,13,14,15
for rep in 1 2; do
for shape in "50432 3072 768" "50432 2304 768"; do
  timeout 120 tools/tn_probe.bin $shape >> $out/tn_probe.log 2>&1
done
done
cat $out/tn_probe.log
