#!/bin/bash
# same-box A/B of two library builds on another bench configuration: bash tools/r06_cfg_ab.sh <tag> <other lib> <bench args...>
export PYTHONPATH=$PWD
tag=$1; other=$2; shift 2
ms() { grep -o '"ms_per_step": [0-9.]*' $1 | head -1; }
for i in 1 2; do
  VITK_LIB=$other timeout 600 python bench.py --no-cpu-baseline "$@" > gpurun_out/${tag}_other_$i.json.log 2>&1; echo "other   $(ms gpurun_out/${tag}_other_$i.json.log)"
  timeout 600 python bench.py --no-cpu-baseline "$@" > gpurun_out/${tag}_this_$i.json.log 2>&1; echo "this    $(ms gpurun_out/${tag}_this_$i.json.log)"
done
