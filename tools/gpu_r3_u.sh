#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 120 python tools/time_tower_f32.py 100 2>&1 | grep -v amdgpu.ids
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "dnn_fused or train_golden or gemm or fused_tower" > gpurun_out/u_tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/u_tests.log
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-configs --no-gather-sweep > gpurun_out/u_bench.json 2> gpurun_out/u_bench.err || { tail -5 gpurun_out/u_bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open("gpurun_out/u_bench.json").read().strip().splitlines()[-1])
print("ms", round(d["ms_per_step"],4), "loss", d["config"]["final_loss"])
PY
done
