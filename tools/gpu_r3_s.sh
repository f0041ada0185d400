#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_tower_x6.py -q -x > gpurun_out/x6.log 2>&1 || { tail -30 gpurun_out/x6.log; exit 1; }
tail -2 gpurun_out/x6.log
timeout -k 10 120 python tools/time_tower_kernels.py 200 2>&1 | grep -v amdgpu.ids
for m in 0 2; do
  timeout -k 10 400 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-gather-sweep --tower-mode $m > gpurun_out/s_bench_$m.json 2> gpurun_out/s_bench_$m.err || { echo "bench mode $m failed"; tail -5 gpurun_out/s_bench_$m.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/s_bench_$m.json").read().strip().splitlines()[-1])
print("tower_mode", $m, "ms", round(d["ms_per_step"],4), [(e["name"][:16], round(e["ms_per_step"],4)) for e in d.get("extra_configs",[])])
PY
done
