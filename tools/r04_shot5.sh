#!/bin/bash
# round 4, GPU call 5: TN probe variants (ring depth 3 / 4 / 5, padded vs unpadded image)
set -u
root=$PWD; out=$root/gpurun_out/r04_shot5; mkdir -p $out
for shape in "50432 3072 768" "50432 768 3072" "50432 2304 768" "50432 768 768" "50007 768 520"; do
  timeout 120 tools/tn_probe.bin $shape >> $out/tn_probe.log 2>&1
done
cat $out/tn_probe.log
