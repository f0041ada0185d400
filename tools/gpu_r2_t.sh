#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
cd $root
timeout -k 10 1000 python3 -m pytest tests/test_gpu_sharded.py -q -x > $out/r2t_pytest.log 2>&1
rc=$?
tail -5 $out/r2t_pytest.log
if [ $rc -ne 0 ]; then grep -E "^E " $out/r2t_pytest.log | head -30; exit 1; fi
