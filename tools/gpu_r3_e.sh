#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "cin or xdeepfm or fused_tower" > gpurun_out/e_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/e_tests.log
tail -6 gpurun_out/e_tests.log
for i in 1 2 3; do timeout -k 10 120 python tools/time_layers.py cin 30 split 2>&1 | grep -v amdgpu; done
