#!/bin/bash
# round 4, GPU call 33: staggered start of the NT GEMM's workgroups (slack-based and forced) -- the eight layer shapes, then the step
set -u
root=$PWD; export PYTHONPATH=$root; out=$root/gpurun_out/r04e; mkdir -p $out
log=$out/r04e_nt_stagger.log; : > $log
for rep in 1 2; do
for st in 0 250 360 450; do
  VITK_NTP_STAGGER=$st timeout 300 python tools/nt_shapes.py 4 >> $log 2>>$out/err.log
done
for f in 20000 40000; do
  VITK_NTP_STAGGER_FORCE=$f timeout 300 python tools/nt_shapes.py 4 >> $log 2>>$out/err.log
done
done
for st in 0 360 0 360; do
  echo "== bench VITK_NTP_STAGGER=$st" >> $log
  VITK_NTP_STAGGER=$st timeout 600 python bench.py --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline 2>>$out/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step_all'])" >> $log
done
cat $log
