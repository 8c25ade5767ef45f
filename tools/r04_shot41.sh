#!/bin/bash
# round 4, GPU call 41: the LDS-DMA varlen attention kernels in the product: NaViT + variants + parity tests, then configs 5 and 4, old library beside the new one
set -u
root=$PWD; export PYTHONPATH=$root; out=$root/gpurun_out/r04g; mkdir -p $out
timeout 900 python -m pytest tests/test_navit_gpu.py tests/test_variants_gpu.py tests/test_fp8_gpu.py tests/test_memory_gpu.py -x -q > $out/r04g_tests_attention.log 2>&1; echo "rc=$?" >> $out/r04g_tests_attention.log; tail -4 $out/r04g_tests_attention.log
log=$out/r04g_cfg45_ab.log; : > $log
for rep in 1 2; do
for lib in libvitk_oldvl.so libvitk.so; do
  echo "== vit_h14 fp8 $lib (rep $rep)" >> $log
  VITK_LIB=$root/vit_pytorch_amd/$lib timeout 600 python bench.py --config vit_h14 --steps 6 --warmup 3 --repeats 2 --no-cpu-baseline 2>>$out/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step_all'])" >> $log
done; done
for lib in libvitk_oldvl.so libvitk.so libvitk_oldvl.so libvitk.so; do
  echo "== navit $lib" >> $log
  VITK_LIB=$root/vit_pytorch_amd/$lib timeout 600 python bench.py --config navit --steps 10 --warmup 3 --repeats 2 --no-cpu-baseline 2>>$out/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step_all'])" >> $log
done
cat $log; tail -3 $out/err.log
