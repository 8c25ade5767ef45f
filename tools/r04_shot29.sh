#!/bin/bash
set -u
root=$PWD; export PYTHONPATH=$root; out=$root/gpurun_out/r04_shot29; mkdir -p $out
for cfg in "VITK_ATTN_FWD_PERSIST=0" "VITK_ATTN_FWD_PERSIST=2 VITK_ATTN_FWD_DELAY=0" "VITK_ATTN_FWD_PERSIST=2 VITK_ATTN_FWD_DELAY=6000" "VITK_ATTN_FWD_PERSIST=2 VITK_ATTN_FWD_DELAY=12000" "VITK_ATTN_FWD_PERSIST=2 VITK_ATTN_FWD_DELAY=20000" "VITK_ATTN_FWD_PERSIST=2 VITK_ATTN_FWD_DELAY=30000" "VITK_ATTN_FWD_PERSIST=0"; do
  env $cfg python tools/attn_fwd_ab.py 2>&1 | grep -v amdgpu | tee -a $out/attn.log
done
