#!/bin/bash
set -u
root=$PWD; out=$root/gpurun_out/r04_shot30; mkdir -p $out
export TN_PROBE_VARS=0,13,14,15
for rep in 1 2; do
for shape in "50432 3072 768" "50432 2304 768"; do
  timeout 120 tools/tn_probe.bin $shape >> $out/tn_probe.log 2>&1
done
done
grep -v "^w128\|^M=" $out/tn_probe.log
