export PYTHONPATH=$PWD
python -m pytest tests -q -x -m gpu 2>&1 | tail -4
python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step_all'], d['roofline'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r01c_stats -o run --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r01c_stats.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/r01c_stats.log | cut -c1-100
grep "gemm_nt256pp_kernel<2" $GRAFT_REPO_ROOT/gpurun_out/r01c_stats/run_kernel_stats.csv | cut -d, -f2-5
