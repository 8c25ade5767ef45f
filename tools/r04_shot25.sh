#!/bin/bash
set -u
root=$PWD; export PYTHONPATH=$root; out=$root/gpurun_out/r04_shot25; mkdir -p $out
for i in 1 2 3; do
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('side stream on ', d['ms_per_step'], d['ms_per_step_all'])" | tee -a $out/ab.log
VITK_DW_STREAM=0 timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('side stream off', d['ms_per_step'], d['ms_per_step_all'])" | tee -a $out/ab.log
done
