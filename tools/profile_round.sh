#!/bin/bash
# Everything profiles/ holds for a round, in one GPU call:  bash tools/profile_round.sh [tag]
#   gpurun_out/<tag>_bench.json.log         default `python bench.py` line (with cpu_baseline)
#   gpurun_out/<tag>_stats_serialized/  rocprofv3 --kernel-trace --stats of a short bench run (bench.py --no-cpu-baseline: the timed workload only)
#   gpurun_out/<tag>_pmc_{fetch,write}/     rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/kprof.py
#   gpurun_out/<tag>_pmc_traffic.json, <tag>_pmc_mfma_util.json, <tag>_kernel_stats_*.csv   the summaries profiles/ keeps
set -u
tag=${1:-r04}
root=$PWD
export PYTHONPATH=$root
out=$root/gpurun_out
mkdir -p $out
timeout 400 python bench.py > $out/${tag}_bench.json.log 2> $out/${tag}_bench.err
tail -1 $out/${tag}_bench.json.log | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $out/${tag}_stats_serialized -o run --output-format csv -- python $root/bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline > $out/${tag}_stats_serialized.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/${tag}_pmc_fetch -o run --output-format csv -- python $root/tools/kprof.py > $out/${tag}_pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/${tag}_pmc_write -o run --output-format csv -- python $root/tools/kprof.py > $out/${tag}_pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES -d $out/${tag}_pmc_mfma -o run --output-format csv -- python $root/tools/kprof.py > $out/${tag}_pmc_mfma.log 2>&1
cd $root
python tools/pmc_traffic_json.py $out/${tag}_pmc_fetch $out/${tag}_pmc_write $out/${tag}_pmc_traffic.json $out/${tag}_pmc_mfma $out/${tag}_pmc_mfma_util.json
for f in serialized; do cp $out/${tag}_stats_$f/*/run_kernel_stats.csv $out/${tag}_kernel_stats_$f.csv 2>/dev/null || cp $out/${tag}_stats_$f/run_kernel_stats.csv $out/${tag}_kernel_stats_$f.csv; done
ls $out/${tag}_stats_serialized $out/${tag}_pmc_fetch | head
