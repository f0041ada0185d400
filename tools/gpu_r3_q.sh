#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_tower_x6.py -q -x > gpurun_out/x6.log 2>&1 || { tail -30 gpurun_out/x6.log; exit 1; }
tail -2 gpurun_out/x6.log
bash tools/gpu_r3_p.sh 2
