#!/bin/bash
# round 4, GPU call 1: the never-run TN 128x128-wave-tile probe at the four dW shapes, the 16-bit forward stream A/B, a baseline bench line
set -u
root=$PWD; export PYTHONPATH=$root; out=$root/gpurun_out/r04_shot1; mkdir -p $out
for shape in "50432 3072 768" "50432 768 3072" "50432 2304 768" "50432 768 768"; do
  timeout 120 tools/tn_wave128.bin $shape >> $out/tn_wave128.log 2>&1
done
cat $out/tn_wave128.log
timeout 300 python tools/step_ab.py "VITK_FWD_STREAM=f32" "VITK_FWD_STREAM=16" --rounds 5 --steps 5 > $out/stream16_ab.log 2>&1
tail -3 $out/stream16_ab.log
timeout 300 python bench.py --no-cpu-baseline > $out/bench.log 2>$out/bench.err
tail -1 $out/bench.log | cut -c1-600
