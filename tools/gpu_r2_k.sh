#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
cd $root
timeout -k 10 900 python3 -m pytest tests/test_gpu_fused_tower.py tests/test_gpu_models_step.py tests/test_gpu_train_golden.py tests/test_gpu_cin.py -q -x > $out/r2k_pytest.log 2>&1
rc=$?
tail -5 $out/r2k_pytest.log
if [ $rc -ne 0 ]; then grep -E "^E " $out/r2k_pytest.log | head -20; exit 1; fi
cd /tmp && export TMPDIR=/tmp
python3 $root/tools/time_models.py 100 xdeepfm 2>&1 | tail -3
