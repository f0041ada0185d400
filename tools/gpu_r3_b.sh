#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -k "train_golden or fullsize or cin_cfg3" -s > gpurun_out/t1.log 2>&1
echo "pytest rc $?" >> gpurun_out/t1.log
grep -E "passed|failed|fullsize|^FAILED|^E  +Assert" gpurun_out/t1.log | tail -40
