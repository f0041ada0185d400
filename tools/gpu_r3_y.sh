#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "attention or fullsize or train_golden or sharded or layernorm or gemm" > gpurun_out/y_tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/y_tests.log
bash tools/ktrace_model.sh attention_deepfm 40 2>&1 | grep -v "^W2026" | head -30
