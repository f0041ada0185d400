#!/bin/bash
mkdir -p gpurun_out
run() {
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-configs --no-gather-sweep "$@" > gpurun_out/x_bench.json 2> gpurun_out/x_bench.err || { tail -8 gpurun_out/x_bench.err; exit 1; }
  python - "$*" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/x_bench.json").read().strip().splitlines()[-1])
r=d["roofline"]
print(f"{sys.argv[1]:40s} ms {d['ms_per_step']:.4f} loss {d['config']['final_loss']:.6f} gather {r['achieved']:.0f} GB/s n_in {r['launches_timed_in_region']}")
PY
}
for i in 1 2 3; do
run --steps 20 --warmup 5
done
run --steps 100 --warmup 20
run --steps 300 --warmup 20
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "packed or fused_tower or train_golden or sharded or dp_rehearsal" > gpurun_out/x_tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/x_tests.log
