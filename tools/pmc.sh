#!/bin/bash
# Per-kernel hardware counters for a driver script: tools/pmc.sh <script.py> <out-name> "<CTR CTR ..>" ["<CTR ..>" ...]
# One rocprofv3 --pmc pass per counter group (pure --pmc + kernel trace: no sys/hip tracing), CSVs under
# gpurun_out/<out-name>/<i>/ ; summarise with tools/pmc_sum.py gpurun_out/<out-name>.
set -u
script=$1; name=$2; shift 2
export PYTHONPATH=$PWD
root=$PWD
out=$root/gpurun_out/$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  rocprofv3 --kernel-trace --pmc $grp -d $out/$i -o run --output-format csv -- python $root/$script > $out/$i.log 2>&1
  i=$((i+1))
done
cd $root
python tools/pmc_sum.py $out
