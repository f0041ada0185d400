#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
export DFM_FORCE_DP_PATH=1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/shard_step -o run -- python3 $root/bench.py --steps 48 --warmup 8 --no-cpu-baseline --no-extra-configs --no-gather-timing > $out/shard_step.log 2>&1
python3 $root/tools/kstats.py $out/shard_step 56 45
