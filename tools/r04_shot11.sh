#!/bin/bash
# round 4, GPU call 11: store turns (per-XCD FIFO semaphore around the NT epilogue) -- the eight ViT-B/16 shapes by concurrency limit
set -u
root=$PWD; export PYTHONPATH=$root; out=$root/gpurun_out/r04_shot11; mkdir -p $out
for t in 0 4 8 12 16 24 0 8; do
  echo "=== VITK_NTP_TURNS=$t" >> $out/times.log
  VITK_NTP_TURNS=$t timeout 120 python tools/nt_shapes.py 3 >> $out/times.log 2>&1
done
grep -v amdgpu.ids $out/times.log
