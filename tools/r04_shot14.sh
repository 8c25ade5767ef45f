#!/bin/bash
# round 4, GPU call 14: driver-checkable lines for BASELINE configs 3, 4, 5 (>= 10 steps x 3 windows) + rocprofv3 kernel stats of each
set -u
root=$PWD; export PYTHONPATH=$root; out=$root/gpurun_out/r04_cfgs; mkdir -p $out
timeout 600 python bench.py --config vit_l16 --steps 10 --warmup 3 --repeats 3 --no-cpu-baseline > $out/r04_vit_l16.json.log 2> $out/vit_l16.err
timeout 900 python bench.py --config vit_h14 --steps 10 --warmup 4 --repeats 3 --no-cpu-baseline > $out/r04_vit_h14_fp8.json.log 2> $out/vit_h14.err
timeout 600 python bench.py --config navit --steps 10 --warmup 3 --repeats 3 > $out/r04_navit.json.log 2> $out/navit.err
for f in vit_l16 vit_h14_fp8 navit; do tail -1 $out/r04_$f.json.log | cut -c1-330; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $out/stats_l16 -o run --output-format csv -- python $root/bench.py --config vit_l16 --steps 4 --warmup 2 --repeats 1 --no-cpu-baseline > $out/stats_l16.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -d $out/stats_h14 -o run --output-format csv -- python $root/bench.py --config vit_h14 --steps 3 --warmup 3 --repeats 1 --no-cpu-baseline > $out/stats_h14.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $out/stats_navit -o run --output-format csv -- python $root/bench.py --config navit --steps 4 --warmup 2 --repeats 1 > $out/stats_navit.log 2>&1
cd $root
for f in l16 h14 navit; do cp $out/stats_$f/*/run_kernel_stats.csv $out/r04_${f}_kernel_stats.csv 2>/dev/null || cp $out/stats_$f/run_kernel_stats.csv $out/r04_${f}_kernel_stats.csv; done
ls -la $out | head -30
