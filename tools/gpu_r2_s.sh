#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
cd $root
timeout -k 10 900 python3 -m pytest tests/test_gpu_attention.py tests/test_gpu_fused_tower.py -q -x > $out/r2s_pytest.log 2>&1
rc=$?
tail -4 $out/r2s_pytest.log
if [ $rc -ne 0 ]; then grep -E "^E " $out/r2s_pytest.log | head -20; exit 1; fi
python3 tools/time_layers.py attn 30
python3 tools/time_models.py 200 attention_deepfm 2>&1 | tail -1
bash tools/gpu_r2_m.sh
