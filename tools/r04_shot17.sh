#!/bin/bash
set -u
root=$PWD; export PYTHONPATH=$root; out=$root/gpurun_out/r04_shot17; mkdir -p $out
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "layernorm" 2>&1 | tail -3 > $out/tests.log; cat $out/tests.log
for r in 1 2; do
VITK_LN_FWD16=0 python tools/ln16_ab.py 2>&1 | grep -v amdgpu >> $out/ln16.log
python tools/ln16_ab.py 2>&1 | grep -v amdgpu >> $out/ln16.log
done
cat $out/ln16.log
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_all'])"
VITK_LN_FWD16=0 timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('off', d['ms_per_step'], d['ms_per_step_all'])"
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_all'])"
