#!/bin/bash
# the round's last GPU seconds: the RESID16 epilogue and the 16-bit forward stream option, then (if time remains) its step time
set -u
export PYTHONPATH=$PWD
out=$PWD/gpurun_out; mkdir -p $out
timeout 24 python -m pytest tests/test_stream16_gpu.py -q -m gpu -s -p no:cacheprovider > $out/last_stream16_tests.log 2>&1
tail -12 $out/last_stream16_tests.log | cut -c1-300
VITK_FWD_STREAM=16 timeout 20 python bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline > $out/last_stream16_bench.log 2>&1
tail -1 $out/last_stream16_bench.log | cut -c1-300
