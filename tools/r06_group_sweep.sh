export PYTHONPATH=$PWD
export VITK_LIB=$PWD/vit_pytorch_amd/libvitk_exp.so
ms() { grep -o '"ms_per_step": [0-9.]*' $1 | head -1; }
for i in 1 2; do for g in default 3 6 12 default; do
  if [ $g = default ]; then unset VITK_GROUP_N; else export VITK_GROUP_N=$g; fi
  timeout 300 python bench.py --no-cpu-baseline --repeats 2 > gpurun_out/r06w_group_${g}_$i.log 2>&1; echo "group_n $g run $i: $(ms gpurun_out/r06w_group_${g}_$i.log)"
done; done
