#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "not fullsize and not auc" > gpurun_out/d_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/d_tests.log
tail -6 gpurun_out/d_tests.log
for i in 1 2; do
for mode in "" "--rowplan-inline"; do
timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-configs --no-gather-sweep $mode > gpurun_out/d_bench.json 2> gpurun_out/d_bench.err || { tail -5 gpurun_out/d_bench.err; exit 1; }
python - "$mode" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/d_bench.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("mode[%s]"%sys.argv[1],"ms",round(d["ms_per_step"],4),"gather avg",round(r["avg_launch_us"],2),"min",round(r["min_launch_us"],2),"n",r["launches_timed"],"frac",round(r["frac"],3), d["config"]["rowplan"][:20], "loss", d["config"]["final_loss"])
PY
done; done
