#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
cd $root
export HSA_ENABLE_IPC_MODE_LEGACY=0
DFM_FORCE_DP_PATH=1 timeout -k 10 300 python3 -X faulthandler bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-extra-configs > $out/r2w.json 2> $out/r2w.err
echo "rc=$?"
wc -c $out/r2w.json
tail -20 $out/r2w.err
