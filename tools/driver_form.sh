#!/bin/bash
# the driver's form (--steps 20 --warmup 5) against variants: where do the us per step over the long run go?
mkdir -p gpurun_out
run() {
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --no-gather-sweep "$@" > gpurun_out/w_bench.json 2> gpurun_out/w_bench.err || { tail -5 gpurun_out/w_bench.err; exit 1; }
  python - "$*" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/w_bench.json").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:50s} ms {d['ms_per_step']:.4f} loss {d['config']['final_loss']:.6f}")
PY
}
for i in 1 2 3 4; do
run
done
run --region-order singles-first
run --no-gather-timing --steps-per-graph 5
