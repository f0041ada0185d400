#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_fused_tower.py -m gpu -q -k "linear_backward_matches_torch" > gpurun_out/n0.log 2>&1; echo "mode0 rc $?"; tail -3 gpurun_out/n0.log
DFM_TEST_TOWER_MODE=1 timeout -k 10 600 python -m pytest tests/test_gpu_fused_tower.py -m gpu -q -k "linear_backward_matches_torch and shape4" > gpurun_out/n1.log 2>&1; echo "mode1 rc $?"; grep -E "^E  |passed|failed" gpurun_out/n1.log | cut -c1-200 | tail -8
