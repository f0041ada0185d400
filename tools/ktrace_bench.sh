#!/bin/bash
# Kernel statistics of the headline step only (GPU box): rocprofv3 kernel trace of a short bench run.
# usage: bash tools/ktrace_bench.sh [extra bench.py arguments ...]
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $out/kt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o run -- python3 $root/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extra-configs --no-gather-timing "$@" > $out/kt.log 2>&1 || { tail -5 $out/kt.log; exit 1; }
python3 $root/tools/kstats.py $out/kt 120 30 | cut -c1-60,97-200
