#!/bin/bash
mkdir -p gpurun_out
for spg in 4; do
timeout -k 10 200 python bench.py --steps 240 --warmup 24 --no-cpu-baseline --no-extra-configs --no-gather-sweep --steps-per-graph $spg > gpurun_out/g_bench.json 2> gpurun_out/g_bench.err || { tail -5 gpurun_out/g_bench.err; exit 1; }
python - "$spg" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/g_bench.json").read().strip().splitlines()[-1])
print("spg",sys.argv[1],"ms",round(d["ms_per_step"],4))
PY
done
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --no-gather-sweep > gpurun_out/g_bench.json 2> gpurun_out/g_bench.err || { tail -5 gpurun_out/g_bench.err; exit 1; }
python - <<'PY'
import json,sys
d=json.loads(open("gpurun_out/g_bench.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("driver-like 20 steps: ms",round(d["ms_per_step"],4),"gather", round(r["avg_launch_us"],2), r["launches_timed_in_region"], r["launches_timed_after_region"])
PY
done
