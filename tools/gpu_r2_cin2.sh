#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
cd $root
timeout -k 10 600 python3 -m pytest tests/test_gpu_cin.py tests/test_gpu_fused_tower.py tests/test_gpu_models_step.py -q -x > $out/cin2_pytest.log 2>&1
rc=$?
tail -3 $out/cin2_pytest.log
if [ $rc -ne 0 ]; then grep -E "^E " $out/cin2_pytest.log | head -30; exit 1; fi
python3 tools/time_layers.py cin 30
python3 tools/time_models.py 100 xdeepfm 2>&1 | tail -1
