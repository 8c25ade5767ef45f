#!/bin/bash
# round 4, GPU call 44 (end-of-round evidence): smoke(), the profile round of the headline (tag r04h), BASELINE configs 3 / 4 / 5 (10 steps x 3 windows + kernel stats)
set -u
root=$PWD; export PYTHONPATH=$root; out=$root/gpurun_out; mkdir -p $out/r04h_cfgs
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $out/r04h_smoke.log 2>&1; echo "smoke rc=$?" >> $out/r04h_smoke.log; tail -4 $out/r04h_smoke.log
timeout 1200 bash tools/profile_round.sh r04h > $out/r04h_profile_round.log 2>&1; tail -3 $out/r04h_profile_round.log
o=$out/r04h_cfgs
timeout 600 python bench.py --config vit_l16 --steps 10 --warmup 3 --repeats 3 --no-cpu-baseline > $o/r04h_vit_l16.json.log 2> $o/vit_l16.err
timeout 900 python bench.py --config vit_h14 --steps 10 --warmup 4 --repeats 3 --no-cpu-baseline > $o/r04h_vit_h14_fp8.json.log 2> $o/vit_h14.err
timeout 600 python bench.py --config navit --steps 10 --warmup 3 --repeats 3 > $o/r04h_navit.json.log 2> $o/navit.err
for f in vit_l16 vit_h14_fp8 navit; do tail -1 $o/r04h_$f.json.log | cut -c1-260; done
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $o/stats_h14 -o run --output-format csv -- python $root/bench.py --config vit_h14 --steps 3 --warmup 3 --repeats 1 --no-cpu-baseline > $o/stats_h14.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $o/stats_navit -o run --output-format csv -- python $root/bench.py --config navit --steps 4 --warmup 2 --repeats 1 > $o/stats_navit.log 2>&1
cd $root
for f in h14 navit; do cp $o/stats_$f/*/run_kernel_stats.csv $o/r04h_${f}_kernel_stats.csv 2>/dev/null || cp $o/stats_$f/run_kernel_stats.csv $o/r04h_${f}_kernel_stats.csv; done
rm -rf $o/stats_h14 $o/stats_navit $out/r04h_stats_two_streams $out/r04h_stats_serialized $out/r04h_pmc_fetch $out/r04h_pmc_write $out/r04h_pmc_mfma
ls $o
