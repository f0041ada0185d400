#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
cd $root
timeout -k 10 900 python3 -m pytest tests/test_gpu_fused_tower.py -q -x > $out/r2n_pytest.log 2>&1
rc=$?
tail -4 $out/r2n_pytest.log
if [ $rc -ne 0 ]; then grep -E "^E " $out/r2n_pytest.log | head -20; exit 1; fi
cd /tmp && export TMPDIR=/tmp
for spg in 1 2 4 8; do
  python3 $root/bench.py --steps 400 --no-cpu-baseline --no-extra-configs --steps-per-graph $spg > $out/bench_n.json 2> $out/bench_n.err || { tail -20 $out/bench_n.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('$out/bench_n.json').read().strip().splitlines()[-1])
print('steps/graph $spg:', round(d['value']/1e6,3), 'M/s', round(d['ms_per_step'],4), 'ms  gather us', d['roofline']['avg_launch_us'], d['roofline']['launches_timed'], d['config']['final_loss'])"
done
time python3 $root/bench.py > $out/bench_n_full.json 2> $out/bench_n_full.err || { tail -20 $out/bench_n_full.err; exit 1; }
python3 -c "
import json
d=json.loads(open('$out/bench_n_full.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['launches_timed'])
for e in d['extra_configs']: print(e['workload'][:20], e['ms_per_step'], e['samples_per_s'], e['roofline'].get('fwd_ms'), e['roofline'].get('fwd_bwd_ms'), e['roofline']['frac'])"
