#!/bin/bash
# kernel trace of the headline step in a given tower mode: per-kernel averages
mode=${1:-2}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/p_mode$mode
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o run -- python3 $root/bench.py --steps 100 --warmup 12 --no-cpu-baseline --no-extra-configs --no-gather-sweep --tower-mode $mode > $out/bench.json 2> $out/trace.err || { tail -5 $out/trace.err; exit 1; }
cp $out/trace/run_kernel_stats.csv $out/kernel_stats.csv
rm -rf $out/trace
python3 - $out/kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    print(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.2f}")
PY
