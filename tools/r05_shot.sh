#!/bin/bash
# One GPU call of round 5:  bash tools/r05_shot.sh <tag> <steps...>   (steps: probe tests_nt tests_all bench profile)
set -u
tag=$1; shift
root=$PWD
export PYTHONPATH=$root
out=$root/gpurun_out
mkdir -p $out
for step in "$@"; do
  case $step in
    probe_quick) timeout 600 tools/nt_probe.bin 3 1 > $out/${tag}_nt_probe_quick.log 2>&1; tail -5 $out/${tag}_nt_probe_quick.log ;;
    dg_ab) for i in 1 2; do VITK_GELU_DG=16 timeout 300 python bench.py --no-cpu-baseline > $out/${tag}_bench_dg16_$i.json.log 2>&1; echo dg16 $(grep -o '"ms_per_step": [0-9.]*' $out/${tag}_bench_dg16_$i.json.log | head -1); timeout 300 python bench.py --no-cpu-baseline > $out/${tag}_bench_dg8_$i.json.log 2>&1; echo dg8 $(grep -o '"ms_per_step": [0-9.]*' $out/${tag}_bench_dg8_$i.json.log | head -1); done ;;
    parity_log) timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -s -x > $out/${tag}_parity_dg8.log 2>&1; tail -2 $out/${tag}_parity_dg8.log; VITK_GELU_DG=16 timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -s -x > $out/${tag}_parity_dg16.log 2>&1; tail -2 $out/${tag}_parity_dg16.log ;;
    probe_instep) timeout 600 tools/nt_probe.bin 3 2 > $out/${tag}_nt_probe_instep.log 2>&1; cat $out/${tag}_nt_probe_instep.log ;;
    r04_ab) for i in 1 2; do (cd _ab_r04 && PYTHONPATH=$PWD timeout 300 python bench.py --no-cpu-baseline > $out/${tag}_bench_r04tree_$i.json.log 2>&1); echo r04tree $(grep -o '"ms_per_step": [0-9.]*' $out/${tag}_bench_r04tree_$i.json.log | head -1); timeout 300 python bench.py --no-cpu-baseline > $out/${tag}_bench_r05tree_$i.json.log 2>&1; echo r05tree $(grep -o '"ms_per_step": [0-9.]*' $out/${tag}_bench_r05tree_$i.json.log | head -1); done ;;
    group_sweep) for i in 1 2; do for g in default 3 4 6 9 12; do if [ $g = default ]; then unset VITK_GROUP_N; else export VITK_GROUP_N=$g; fi; timeout 300 python bench.py --no-cpu-baseline --repeats 2 > $out/${tag}_bench_group${g}_$i.json.log 2>&1; echo group_n $g $(grep -o '"ms_per_step": [0-9.]*' $out/${tag}_bench_group${g}_$i.json.log | head -1); done; unset VITK_GROUP_N; for sp in r n; do VITK_NTW_SPLIT=$sp timeout 300 python bench.py --no-cpu-baseline --repeats 2 > $out/${tag}_bench_split${sp}_$i.json.log 2>&1; echo split $sp $(grep -o '"ms_per_step": [0-9.]*' $out/${tag}_bench_split${sp}_$i.json.log | head -1); done; done ;;   # needs the VITK_BUILD_EXPERIMENTS=1 library
    group_confirm) for i in 1 2 3; do for v in "default::" "group4:4:" "splitn::n" "both:4:n" "group3:3:" "g3n:3:n"; do n=${v%%:*}; r=${v#*:}; g=${r%%:*}; sp=${r#*:}; if [ -z "$g" ]; then unset VITK_GROUP_N; else export VITK_GROUP_N=$g; fi; if [ -z "$sp" ]; then unset VITK_NTW_SPLIT; else export VITK_NTW_SPLIT=$sp; fi; timeout 300 python bench.py --no-cpu-baseline --repeats 2 > $out/${tag}_bench_${n}_$i.json.log 2>&1; echo $n $(grep -o '"ms_per_step": [0-9.]*' $out/${tag}_bench_${n}_$i.json.log | head -1); done; done; unset VITK_GROUP_N VITK_NTW_SPLIT ;;   # needs the VITK_BUILD_EXPERIMENTS=1 library
    probe) timeout 600 tools/nt_probe.bin 3 > $out/${tag}_nt_probe.log 2>&1; tail -5 $out/${tag}_nt_probe.log ;;
    tests_nt) timeout 900 python -m pytest tests/test_gemm_nt_w128_gpu.py tests/test_gemm_persist_gpu.py tests/test_kernels_gpu.py tests/test_headline_extents_gpu.py tests/test_fuzz_ops_gpu.py -m gpu -x -q > $out/${tag}_tests_nt.log 2>&1; tail -3 $out/${tag}_tests_nt.log ;;
    tests_all) timeout 2400 python -m pytest tests -m gpu -x -q > $out/${tag}_tests_all.log 2>&1; tail -3 $out/${tag}_tests_all.log ;;
    bench) timeout 400 python bench.py > $out/${tag}_bench.json.log 2> $out/${tag}_bench.err; tail -1 $out/${tag}_bench.json.log | cut -c1-600 ;;
    bench_ab) for i in 1 2; do VITK_NT_W128=0 timeout 300 python bench.py --no-cpu-baseline > $out/${tag}_bench_old_$i.json.log 2>&1; timeout 300 python bench.py --no-cpu-baseline > $out/${tag}_bench_new_$i.json.log 2>&1; done; grep -h -o '"ms_per_step": [0-9.]*' $out/${tag}_bench_old_*.json.log $out/${tag}_bench_new_*.json.log ;;
    tests_new) timeout 1200 python -m pytest tests/test_gemm_nt_w128_gpu.py tests/test_gemm_persist_gpu.py tests/test_kernels_gpu.py tests/test_navit_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q > $out/${tag}_tests_new.log 2>&1; tail -3 $out/${tag}_tests_new.log ;;
    bench_ab4) for v in "default::" "r05d:rounds:all" "split_rounds:rounds:" "relax_all::all" "default2::"; do n=${v%%:*}; r=${v#*:}; sp=${r%%:*}; rx=${r#*:}; VITK_NTW_SPLIT=$sp VITK_NTW_RELAX=$rx timeout 300 python bench.py --no-cpu-baseline > $out/${tag}_bench_$n.json.log 2>&1; echo $n $(grep -o '"ms_per_step": [0-9.]*' $out/${tag}_bench_$n.json.log | head -1); done ;;
    tree_ab) for i in 1 2; do (cd _ab_r05d && PYTHONPATH=$PWD timeout 300 python bench.py --no-cpu-baseline > $out/${tag}_bench_oldtree_$i.json.log 2>&1); echo oldtree $(grep -o '"ms_per_step": [0-9.]*' $out/${tag}_bench_oldtree_$i.json.log | head -1); timeout 300 python bench.py --no-cpu-baseline > $out/${tag}_bench_newtree_$i.json.log 2>&1; echo newtree $(grep -o '"ms_per_step": [0-9.]*' $out/${tag}_bench_newtree_$i.json.log | head -1); done ;;
    fold_ab) for i in 1 2; do VITK_FOLD_DEFER=0 timeout 300 python bench.py --no-cpu-baseline > $out/${tag}_bench_nofold_$i.json.log 2>&1; echo nofold $(grep -o '"ms_per_step": [0-9.]*' $out/${tag}_bench_nofold_$i.json.log | head -1); timeout 300 python bench.py --no-cpu-baseline > $out/${tag}_bench_fold_$i.json.log 2>&1; echo fold $(grep -o '"ms_per_step": [0-9.]*' $out/${tag}_bench_fold_$i.json.log | head -1); done ;;
    mask_ab) for i in 1 2; do for m in 0 0x23 0xff; do VITK_NT_W128=$m timeout 300 python bench.py --no-cpu-baseline > $out/${tag}_bench_mask${m}_$i.json.log 2>&1; echo mask $m $(grep -o '"ms_per_step": [0-9.]*' $out/${tag}_bench_mask${m}_$i.json.log | head -1); done; done ;;
    profile) bash tools/profile_round.sh $tag ;;
  esac
done
