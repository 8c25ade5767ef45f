#!/bin/bash
set -u
root=$PWD; export PYTHONPATH=$root; out=$root/gpurun_out/r04_shot24; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_abi.py tests/test_parity_gpu.py tests/test_parallel_gpu.py -q -m gpu -x -k "tn or abi or production or parallel or full or dropin" 2>&1 | tail -5 > $out/tests.log; cat $out/tests.log
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pair on ', d['ms_per_step'], d['ms_per_step_all'], [(c['class'], c['ms_per_step']) for c in d['roofline']['classes']])" | tee -a $out/ab.log
VITK_TN_PAIR=0 timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pair off', d['ms_per_step'], d['ms_per_step_all'], [(c['class'], c['ms_per_step']) for c in d['roofline']['classes']])" | tee -a $out/ab.log
done
