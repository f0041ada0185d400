#!/bin/bash
# Runs on the GPU box after tools/profile_round.sh: kernel statistics of the CIN and attention layers
# (BASELINE.json configurations 3 / 4, layer forward + backward in isolation), of the field-sharded step
# with one rank over RCCL, of one rank of a simulated 8-rank job, and the replicated layout's optimizer
# tail.  Output under gpurun_out/<tag>/; tools/collect_profiles.py copies it to profiles/.
# usage: tools/profile_extras.sh <tag>
set -o pipefail
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
for layer in cin attn; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/layer_$layer -o run -- python3 $root/tools/time_layers.py $layer 30 > $out/layer_$layer.txt 2> $out/layer_$layer.err || { tail -5 $out/layer_$layer.err; exit 1; }
  cp $out/layer_$layer/run_kernel_stats.csv $out/layer_${layer}_kernel_stats.csv
  tail -2 $out/layer_$layer.txt
done
DFM_FORCE_DP_PATH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/sharded -o run -- python3 $root/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extra-configs --no-gather-timing > $out/sharded_bench.json 2> $out/sharded.err || { tail -5 $out/sharded.err; exit 1; }
cp $out/sharded/run_kernel_stats.csv $out/sharded_dp1_kernel_stats.csv
DFM_FORCE_DP_PATH=1 python3 $root/bench.py --steps 200 --warmup 40 --no-cpu-baseline --no-extra-configs > $out/sharded_bench_untraced.json 2> $out/sharded2.err || { tail -5 $out/sharded2.err; exit 1; }
DFM_FORCE_DP_PATH=1 python3 $root/bench.py --steps 200 --warmup 40 --no-cpu-baseline --no-extra-configs --dp-mode replicated > $out/replicated_bench_untraced.json 2> $out/repl.err || { tail -5 $out/repl.err; exit 1; }
for w in 1 8; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/sim$w -o run -- python3 $root/tools/time_sharded_sim.py $w 0 30 > $out/sharded_sim$w.txt 2> $out/sim$w.err || { tail -5 $out/sim$w.err; exit 1; }
  python3 $root/tools/kstats.py $out/sim$w 35 60 >> $out/sharded_sim$w.txt
done
python3 $root/tools/time_merge.py 1 2 4 8 > $out/replicated_tail.txt 2> $out/merge.err || { tail -5 $out/merge.err; exit 1; }
cat $out/replicated_tail.txt
