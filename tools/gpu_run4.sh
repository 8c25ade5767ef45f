#!/bin/bash
export PYTHONPATH=$PWD
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r2d_tests.log; grep -v amdgpu.ids gpurun_out/r2d_tests.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r2d_bench.log 2>&1; tail -1 gpurun_out/r2d_bench.log | cut -c1-900
VITK_NTP_EPIS=0 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r2d_bench_old.log 2>&1; tail -1 gpurun_out/r2d_bench_old.log | cut -c1-300
VITK_NTP_EPIS=31 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r2d_bench_all.log 2>&1; tail -1 gpurun_out/r2d_bench_all.log | cut -c1-300
