#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "sharded or dp_rehearsal or checkpoint" > gpurun_out/i_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/i_tests.log
tail -12 gpurun_out/i_tests.log
