#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "fused_tower and graph_node" > gpurun_out/h_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/h_tests.log
tail -5 gpurun_out/h_tests.log
for i in 1 2; do
timeout -k 10 200 python bench.py --steps 240 --warmup 20 --no-cpu-baseline --no-extra-configs --no-gather-sweep > gpurun_out/h_bench.json 2> gpurun_out/h_bench.err || { tail -5 gpurun_out/h_bench.err; exit 1; }
python - <<'PY'
import json,sys
d=json.loads(open("gpurun_out/h_bench.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("long: ms",round(d["ms_per_step"],4),"G",d["config"]["steps_per_graph"],"gather", round(r["avg_launch_us"],2), round(r["min_launch_us"],2), r["launches_timed_in_region"], r["launches_timed_after_region"], "loss", d["config"]["final_loss"])
PY
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --no-gather-sweep > gpurun_out/h_bench.json 2> gpurun_out/h_bench.err || { tail -5 gpurun_out/h_bench.err; exit 1; }
python - <<'PY'
import json,sys
d=json.loads(open("gpurun_out/h_bench.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("driver-like 20 steps: ms",round(d["ms_per_step"],4),"G",d["config"]["steps_per_graph"],"gather", round(r["avg_launch_us"],2), r["launches_timed_in_region"], r["launches_timed_after_region"], "frac", round(r["frac"],3))
PY
done
