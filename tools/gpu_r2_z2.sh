#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
cd $root
timeout -k 10 600 python3 -m pytest tests/test_gpu_sharded.py tests/test_gpu_models_step.py tests/test_gpu_dp_rehearsal.py -q -x > $out/r2z_pytest.log 2>&1
rc=$?
tail -3 $out/r2z_pytest.log
if [ $rc -ne 0 ]; then grep -E "^E " $out/r2z_pytest.log | head -30; exit 1; fi
bash tools/gpu_r2_z.sh
