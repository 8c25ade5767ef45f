#!/bin/bash
# round 4, GPU call 9: everything profiles/ holds for the round (bench line, rocprofv3 kernel stats, PMC traffic / MFMA busy) + tile-order knobs in the step
set -u
root=$PWD; export PYTHONPATH=$root; out=$root/gpurun_out
bash tools/profile_round.sh r04 > $out/r04_profile_round.log 2>&1
tail -25 $out/r04_profile_round.log
for cfg in "X=0" "VITK_NTP_TAIL_LAST=1" "VITK_GROUP_N=4" "VITK_NTP_TAIL_LAST=1 VITK_GROUP_N=4" "X=0" "VITK_NTP_TAIL_LAST=1"; do
  echo "=== $cfg" >> $out/r04_knobs.log
  env $cfg timeout 200 python bench.py --no-cpu-baseline --steps 20 --repeats 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_all'], [(c['class'], c['ms_per_step']) for c in d['roofline']['classes']])" >> $out/r04_knobs.log 2>&1
done
cat $out/r04_knobs.log
