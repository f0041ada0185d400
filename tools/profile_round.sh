#!/bin/bash
# Runs on the GPU box (gpurun): the headline bench (+ configurations 3 and 4), its rocprofv3 kernel trace and
# separate PMC passes for the embedding gather.  Everything lands under gpurun_out/<tag>/; copy what is
# judged into profiles/ (tools/collect_profiles.py).   usage: tools/profile_round.sh <tag>
set -o pipefail
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $root/bench.py --steps 300 --warmup 20 > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
# the driver's own command (defaults), traced
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o run -- python3 $root/bench.py --no-cpu-baseline > $out/bench_traced.json 2> $out/trace.err || { tail -5 $out/trace.err; exit 1; }
cp $out/trace/run_kernel_stats.csv $out/bench_kernel_stats.csv
python3 $root/tools/ktrace_groups.py $out/trace emb_fwd > $out/gather_trace_groups.csv
# PMC: one counter per pass, headline model only
for c in TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_sum; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -o run -- python3 $root/bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-extra-configs > /dev/null 2> $out/pmc_$c.err || { echo "pmc $c failed"; tail -3 $out/pmc_$c.err; continue; }
  python3 $root/tools/pmc_summary.py $out/pmc_$c $c emb_fwd > $out/pmc_$c.txt
  rm -rf $out/pmc_$c
  cat $out/pmc_$c.txt
done
tail -c 1500 $out/bench.json
