#!/bin/bash
# round 3: whole GPU suite with the tower on the bf16 x 6 path (acceptance: nothing fails, no tolerance edited), then
# the headline A/B mode 0 / mode 2 (twice, interleaved)
mkdir -p gpurun_out
DFM_TEST_TOWER_MODE=2 timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/full_m2.log 2>&1
echo "pytest mode 2 rc $?"; tail -25 gpurun_out/full_m2.log
for m in 0 2 0 2; do
  timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-configs --no-gather-sweep --tower-mode $m > gpurun_out/o_bench_$m.json 2> gpurun_out/o_bench_$m.err || { echo "bench mode $m failed"; tail -5 gpurun_out/o_bench_$m.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/o_bench_$m.json").read().strip().splitlines()[-1])
print("tower_mode", $m, "ms", round(d["ms_per_step"],4), "planes", d["config"].get("tower_planes"))
PY
done
