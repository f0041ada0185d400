#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
for w in 1 8; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/shard_sim$w -o run -- python3 $root/tools/time_sharded_sim.py $w 0 30 > $out/shard_sim$w.log 2>&1 || { tail -20 $out/shard_sim$w.log; exit 1; }
tail -1 $out/shard_sim$w.log
python3 $root/tools/kstats.py $out/shard_sim$w 35 60 > $out/shard_sim$w.txt
done
