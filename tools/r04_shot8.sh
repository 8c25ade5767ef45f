#!/bin/bash
# round 4, GPU call 8: NT GEMM tile-order knobs vs time and fabric reads (FETCH_SIZE), the eight ViT-B/16 shapes
set -u
root=$PWD; export PYTHONPATH=$root; out=$root/gpurun_out/r04_shot8; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for cfg in "X=0" "VITK_NTP_TAIL=-1" "VITK_NTP_TAIL_LAST=1" "VITK_GROUP_N=12" "VITK_GROUP_N=4" "VITK_GROUP_N=3" "VITK_GROUP_N=2" "VITK_GROUP_N=12 VITK_NTP_TAIL=-1"; do
  echo "=== $cfg" >> $out/times.log
  (cd $root && env $cfg timeout 120 python tools/nt_shapes.py 3 >> $out/times.log 2>&1)
  (cd $root && env $cfg timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc_$i -o run --output-format csv -- python tools/kprof.py > $out/pmc_$i.log 2>&1)
  echo "=== $cfg" >> $out/fetch.log
  (cd $root && python tools/pmc_sum.py $out/pmc_$i gemm_ntp >> $out/fetch.log 2>&1)
  i=$((i+1))
done
cat $out/times.log
grep -v "^ *$" $out/fetch.log | head -150
