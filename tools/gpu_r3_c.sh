#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/full.log 2>&1
echo "pytest rc $?" >> gpurun_out/full.log
tail -8 gpurun_out/full.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/bench20.json 2> gpurun_out/bench20.err
echo "bench rc $?"
tail -3 gpurun_out/bench20.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench20.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("value",d["value"],"ms",d["ms_per_step"])
print("gather avg",r["avg_launch_us"],"min",r["min_launch_us"],"n",r["launches_timed"],r["launches_timed_in_region"],r["launches_timed_after_region"],"frac",r["frac"],"frac_traffic",r.get("frac_traffic"))
print("sweep",[(x["batch"],round(x["avg_launch_us"],2),round(x["frac"],3)) for x in (r["sweep"] or [])])
for e in d.get("extra_configs",[]): print(e["workload"][:30], e["ms_per_step"], e.get("roofline",{}).get("fwd_ms"), e.get("roofline",{}).get("fwd_bwd_ms"))
print("cpu", d.get("cpu_baseline"))
PY
