#!/bin/bash
set -u
root=$PWD; export PYTHONPATH=$root; out=$root/gpurun_out/r04_shot20; mkdir -p $out
for n in 1 2 3 1 2 4; do
  VITK_DW_STREAMS=$n timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $n', d['ms_per_step'], d['ms_per_step_all'])" | tee -a $out/streams.log
done
