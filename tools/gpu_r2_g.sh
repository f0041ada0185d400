#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $root/bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-gather-timing > $out/bench_g0.json 2> $out/bench_g0.err || { tail -5 $out/bench_g0.err; exit 1; }
DFM_EXP_GATHER_IN_GRAPH=1 python3 $root/bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-gather-timing > $out/bench_g1.json 2> $out/bench_g1.err || { tail -5 $out/bench_g1.err; exit 1; }
for f in g0 g1; do python3 -c "import json,sys; d=json.loads(open('$out/bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"; done
DFM_EXP_GATHER_IN_GRAPH=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/bench_g1t -o run -- python3 $root/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-gather-timing > /dev/null 2> $out/bench_g1t.err
