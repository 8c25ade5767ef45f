#!/bin/bash
# same-box A/B of two builds of the library: bash tools/r06_lib_ab.sh <tag> <path of the other libvitk*.so>   (two interleaved pairs)
export PYTHONPATH=$PWD
tag=$1; other=$2
ms() { grep -o '"ms_per_step": [0-9.]*' $1 | head -1; }
for i in 1 2 3; do
  VITK_LIB=$other timeout 300 python bench.py --no-cpu-baseline > gpurun_out/${tag}_bench_other_$i.json.log 2>&1; echo "other   $(ms gpurun_out/${tag}_bench_other_$i.json.log)"
  timeout 300 python bench.py --no-cpu-baseline > gpurun_out/${tag}_bench_this_$i.json.log 2>&1; echo "this    $(ms gpurun_out/${tag}_bench_this_$i.json.log)"
done
