#!/bin/bash
set -u
root=$PWD; export PYTHONPATH=$root; out=$root/gpurun_out/r04_shot18; mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_abi.py tests/test_parity_gpu.py tests/test_consumers.py tests/test_variants_gpu.py -q -m gpu -x 2>&1 | tail -6 > $out/tests.log; cat $out/tests.log
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_all'])"
VITK_NO_PATCH_LN=1 timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no-gather', d['ms_per_step'], d['ms_per_step_all'])"
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_all'])"
