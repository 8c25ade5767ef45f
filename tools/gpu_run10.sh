#!/bin/bash
export PYTHONPATH=$PWD
for pipe in 1 0 1 0; do echo "PIPE=$pipe"; VITK_NTP_PIPE=$pipe timeout 300 python bench.py --no-cpu-baseline --repeats 2 2>&1 | tail -1 | cut -c1-220; done
