#!/bin/bash
# round 4, GPU call 32: ViT-H/14 fp8 -- weight gradients on the side stream (new fp8 default) vs in line; ViT-L/16 the same pair
set -u
root=$PWD; export PYTHONPATH=$root; out=$root/gpurun_out/r04e; mkdir -p $out
log=$out/r04e_h14_dw_stream_ab.log; : > $log
for rep in 1 2; do
for v in 1 0; do
  echo "== vit_h14 fp8 VITK_DW_STREAM=$v (rep $rep)" >> $log
  VITK_DW_STREAM=$v timeout 600 python bench.py --config vit_h14 --steps 6 --warmup 4 --repeats 2 --no-cpu-baseline 2>>$out/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step_all'], d['memory'])" >> $log
done; done
for v in 0 1; do
  echo "== vit_l16 bf16 VITK_DW_STREAM=$v" >> $log
  VITK_DW_STREAM=$v timeout 600 python bench.py --config vit_l16 --steps 10 --warmup 3 --repeats 2 --no-cpu-baseline 2>>$out/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step_all'])" >> $log
done
cat $log
