#!/bin/bash
# round 4, GPU call 45: fp8 delayed quantiser with 16 elements per thread and trip: fp8 tests + config 5
set -u
root=$PWD; export PYTHONPATH=$root; out=$root/gpurun_out/r04i; mkdir -p $out
timeout 900 python -m pytest tests/test_fp8_backward_gpu.py tests/test_fp8_forward_gpu.py tests/test_fp8_gpu.py -x -q 2>&1 | tail -3 > $out/r04i_fp8_tests.log; cat $out/r04i_fp8_tests.log
timeout 600 python bench.py --config vit_h14 --steps 8 --warmup 3 --repeats 3 --no-cpu-baseline 2>>$out/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step_all'])" > $out/r04i_h14.log; cat $out/r04i_h14.log
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $out/stats_h14 -o run --output-format csv -- python $root/bench.py --config vit_h14 --steps 3 --warmup 3 --repeats 1 --no-cpu-baseline > $out/stats_h14.log 2>&1
cd $root; cp $out/stats_h14/*/run_kernel_stats.csv $out/r04i_h14_kernel_stats.csv 2>/dev/null || cp $out/stats_h14/run_kernel_stats.csv $out/r04i_h14_kernel_stats.csv; rm -rf $out/stats_h14
grep -i "quantize" $out/r04i_h14_kernel_stats.csv | cut -c1-200
