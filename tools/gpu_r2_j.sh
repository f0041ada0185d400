#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
cd $root
timeout -k 10 600 python3 -m pytest tests/test_gpu_fused_tower.py -q -x > $out/r2j_pytest.log 2>&1
tail -3 $out/r2j_pytest.log
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for cfg in "--timing-stride 1" "--timing-stride 8" "--no-gather-timing"; do
  python3 $root/bench.py --steps 400 --no-extra-configs --no-cpu-baseline $cfg > $out/bench_j.json 2> $out/bench_j.err || { tail -20 $out/bench_j.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('$out/bench_j.json').read().strip().splitlines()[-1])
print('$cfg', round(d['value']/1e6,3), 'M/s', round(d['ms_per_step'],4), 'ms  gather us', d['roofline']['avg_launch_us'], d['roofline']['launches_timed'])"
done
done
