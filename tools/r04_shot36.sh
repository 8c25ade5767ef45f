#!/bin/bash
# round 4, GPU call 36: hardware counters of the varlen attention kernels (config-5 shape), geometry 1 and 2, e0 build
set -u
root=$PWD; export PYTHONPATH=$root
export VITK_LIB=$root/vit_pytorch_amd/libvitk_e0.so
for g in 1 2; do
VITK_ATTN_VL=$g timeout 600 bash tools/pmc.sh tools/vl_prof.py r04f_pmc_vl$g "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_SALU" > gpurun_out/r04f_pmc_vl$g.txt 2>&1
tail -70 gpurun_out/r04f_pmc_vl$g.txt
done
