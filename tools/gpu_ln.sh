#!/bin/bash
export PYTHONPATH=$PWD
python tools/ln_ab.py
python tools/kbench.py --ln-wide
python -m pytest tests/test_kernels_gpu.py -q -x -k "layernorm or ln_ or drop" 2>&1 | tail -3
python -m pytest tests/test_parity_gpu.py tests/test_dropout_gpu.py -q -x 2>&1 | tail -3
python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
