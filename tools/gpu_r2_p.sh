#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for sh in 5 6 4; do
  python3 $root/bench.py --steps 400 --no-cpu-baseline --no-extra-configs --gather-shape $sh > $out/bench_p.json 2> $out/bench_p.err || { tail -20 $out/bench_p.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('$out/bench_p.json').read().strip().splitlines()[-1])
print('shape $sh:', round(d['value']/1e6,3), 'M/s', round(d['ms_per_step'],4), 'ms  gather us', round(d['roofline']['avg_launch_us'],2), round(d['roofline']['min_launch_us'],2), d['roofline']['launches_timed'])"
done; done
