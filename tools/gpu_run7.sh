#!/bin/bash
export PYTHONPATH=$PWD
for d in 0 1 2 3; do echo "== VITK_TN_DBG=$d (1 = no DMA in the loop, 2 = no MFMA)"; VITK_TN_DBG=$d timeout 300 python tools/tn_ab.py 2 2>&1 | grep -v amdgpu | cut -c1-110; done
