#!/bin/bash
# One GPU call of round 6:  bash tools/r06_shot.sh <tag> <steps...>
set -u
tag=$1; shift
root=$PWD
export PYTHONPATH=$root
out=$root/gpurun_out
mkdir -p $out
ms() { grep -o '"ms_per_step": [0-9.]*' $1 | head -1; }
for step in "$@"; do
  case $step in
    # env_ab:NAME=VALUE  -- two interleaved pairs of bench runs, default vs the variable set
    env_ab:*) v=${step#env_ab:}; n=$(echo $v | tr "=" "_"); for i in 1 2; do
        env $v timeout 300 python bench.py --no-cpu-baseline > $out/${tag}_bench_${n}_$i.json.log 2>&1; echo "$v $(ms $out/${tag}_bench_${n}_$i.json.log)"
        timeout 300 python bench.py --no-cpu-baseline > $out/${tag}_bench_default_${n}_$i.json.log 2>&1; echo "default $(ms $out/${tag}_bench_default_${n}_$i.json.log)"; done ;;
    probe) timeout 900 tools/nt_probe.bin 3 > $out/${tag}_nt_probe.log 2>&1; tail -3 $out/${tag}_nt_probe.log ;;
    probe_quick) timeout 600 tools/nt_probe.bin 3 1 > $out/${tag}_nt_probe_quick.log 2>&1; tail -3 $out/${tag}_nt_probe_quick.log ;;
    tests_nt) timeout 1200 python -m pytest tests/test_gemm_nt_w128_gpu.py tests/test_gemm_persist_gpu.py tests/test_kernels_gpu.py tests/test_headline_extents_gpu.py tests/test_fuzz_ops_gpu.py -m gpu -x -q > $out/${tag}_tests_nt.log 2>&1; tail -3 $out/${tag}_tests_nt.log ;;
    tests_all) timeout 3000 python -m pytest tests -m gpu -x -q > $out/${tag}_tests_all.log 2>&1; tail -3 $out/${tag}_tests_all.log ;;
    bench) timeout 400 python bench.py > $out/${tag}_bench.json.log 2> $out/${tag}_bench.err; tail -1 $out/${tag}_bench.json.log | cut -c1-400 ;;
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $out/${tag}_smoke.log 2>&1; tail -3 $out/${tag}_smoke.log ;;
    profile) bash tools/profile_round.sh $tag ;;
    *) echo "unknown step $step" ;;
  esac
done
