#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
cd $root
timeout -k 10 900 python3 -m pytest tests/test_gpu_fused_tower.py tests/test_gpu_attention.py -q -x > $out/r2l_pytest.log 2>&1
rc=$?
tail -5 $out/r2l_pytest.log
if [ $rc -ne 0 ]; then grep -E "^E " $out/r2l_pytest.log | head -20; exit 1; fi
cd /tmp && export TMPDIR=/tmp
python3 $root/tools/time_models.py 100 attention_deepfm xdeepfm deepfm 2>&1 | tail -4
