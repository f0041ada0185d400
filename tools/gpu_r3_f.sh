#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "not auc and not sharded and not dp_rehearsal" > gpurun_out/f_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/f_tests.log
tail -5 gpurun_out/f_tests.log
for i in 1 2; do
for mode in "" "--no-plan-lookahead"; do
timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-configs --no-gather-sweep $mode > gpurun_out/f_bench.json 2> gpurun_out/f_bench.err || { tail -5 gpurun_out/f_bench.err; exit 1; }
python - "$mode" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/f_bench.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("mode[%s]"%sys.argv[1],"ms",round(d["ms_per_step"],4),"gather avg",round(r["avg_launch_us"],2),"n",r["launches_timed"],"loss", d["config"]["final_loss"], d["config"]["rowplan"][:40])
PY
done; done
