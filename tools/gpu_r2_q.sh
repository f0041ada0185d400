#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
cd $root
python3 -c "import __graft_entry__ as g; g.smoke()" || exit 1
cd /tmp && export TMPDIR=/tmp
for a in "--steps 20 --warmup 5" "--steps 7 --warmup 0 --no-extra-configs --no-cpu-baseline" "--steps 30 --warmup 3 --no-graph --no-extra-configs --no-cpu-baseline" "--steps 24 --warmup 4 --autograd --no-extra-configs --no-cpu-baseline" "--steps 24 --warmup 4 --unpacked --no-extra-configs --no-cpu-baseline"; do
  python3 $root/bench.py $a > $out/bench_q.json 2> $out/bench_q.err || { echo "FAILED: $a"; tail -20 $out/bench_q.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('$out/bench_q.json').read().strip().splitlines()[-1])
print('$a ->', round(d['value']/1e6,3), 'M/s', round(d['ms_per_step'],4), 'ms gather', d['roofline']['avg_launch_us'], d['roofline']['launches_timed'], d['config']['final_loss'], d['config']['step'], len(d.get('extra_configs',[])))"
done
