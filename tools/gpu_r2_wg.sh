#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
timeout -k 10 120 ./tools/microbench_cin_wgrad || exit 1
timeout -k 10 300 python3 -m pytest tests/test_gpu_cin.py -q -x 2>&1 | tail -3
