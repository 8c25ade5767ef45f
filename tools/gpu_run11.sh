#!/bin/bash
export PYTHONPATH=$PWD
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 > gpurun_out/r2h_tests.log; grep -v amdgpu.ids gpurun_out/r2h_tests.log
timeout 900 python bench.py --config vit_h14 --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline > gpurun_out/r2h_bench_h14.log 2>&1; tail -2 gpurun_out/r2h_bench_h14.log | cut -c1-700
