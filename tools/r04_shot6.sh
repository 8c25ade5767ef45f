#!/bin/bash
# round 4, GPU call 6: HBM/fabric traffic and L2 hit rate of the w128 TN kernel (dW1 shape), separate --pmc passes
set -u
root=$PWD; out=$root/gpurun_out/r04_shot6; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export TN_PROBE_VARS=0,4
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES"; do
  (cd $root && timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $out/$i -o run --output-format csv -- tools/tn_probe.bin 50432 3072 768 > $out/$i.log 2>&1)
  i=$((i+1))
done
cd $root
python tools/pmc_sum.py $out tn_w128 --json $out/pmc.json
