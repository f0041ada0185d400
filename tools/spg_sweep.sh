#!/bin/bash
# Headline step vs steps per graph launch, with and without the gather's timed single steps (GPU box).
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
for a in "--steps-per-graph 4" "--steps-per-graph 8" "--steps-per-graph 4 --no-gather-timing" "--steps-per-graph 8 --no-gather-timing" "--steps-per-graph 16 --no-gather-timing"; do
  python3 bench.py --steps 320 --warmup 32 --no-cpu-baseline --no-extra-configs $a 2>/dev/null > gpurun_out/spg.json || exit 1
  python3 -c "
import json
d=json.loads(open('gpurun_out/spg.json').read().strip().splitlines()[-1])
print('$a', round(d['ms_per_step'],5), d['roofline'].get('frac'))"
done
