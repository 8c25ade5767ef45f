#!/bin/bash
set -u
root=$PWD; export PYTHONPATH=$root; out=$root/gpurun_out/r04_shot28; mkdir -p $out
for i in 1 2; do
for cfg in "X=0" "VITK_NTP_TAIL_LAST=1" "VITK_GROUP_N=4" "VITK_NTP_TAIL_LAST=1 VITK_GROUP_N=4"; do
  env $cfg timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'], d['ms_per_step_all'], [(c['class'], c['ms_per_step']) for c in d['roofline']['classes']])" | tee -a $out/knobs.log
done
done
