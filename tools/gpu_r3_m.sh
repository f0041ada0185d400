#!/bin/bash
mkdir -p gpurun_out
DFM_TEST_TOWER_MODE=1 timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/m_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/m_tests.log
grep -E "^FAILED|passed|failed|rc " gpurun_out/m_tests.log | tail -30
