#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
cd $root
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python3 -m pytest tests/test_gpu_sharded.py tests/test_gpu_dp_rehearsal.py tests/test_gpu_fused_tower.py -q -x > $out/r2y_pytest.log 2>&1
rc=$?
tail -4 $out/r2y_pytest.log
if [ $rc -ne 0 ]; then grep -E "^E " $out/r2y_pytest.log | head -30; exit 1; fi
DFM_FORCE_DP_PATH=1 timeout -k 10 300 python3 bench.py --steps 200 --warmup 40 --no-cpu-baseline --no-extra-configs > $out/r2y_sharded.json 2> $out/r2y_sharded.err || { tail -5 $out/r2y_sharded.err; exit 1; }
python3 -c "
import json
d=json.loads([l for l in open('$out/r2y_sharded.json') if l.startswith('{')][-1]);print(d['ms_per_step'], d['value'], d['config']['final_loss'], d['config']['parallelism'])"
