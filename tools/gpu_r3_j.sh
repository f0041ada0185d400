#!/bin/bash
mkdir -p gpurun_out
for prof in "--vocab-profile criteo" "--vocab 3000000" "--ids zipf"; do
for mode in "" "--no-plan-lookahead"; do
timeout -k 10 300 python bench.py --steps 120 --warmup 16 --no-cpu-baseline --no-extra-configs --no-gather-sweep $prof $mode > gpurun_out/j_bench.json 2> gpurun_out/j_bench.err || { tail -5 gpurun_out/j_bench.err; exit 1; }
python - "$prof $mode" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/j_bench.json").read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[1].ljust(45),"ms",round(d["ms_per_step"],4),"loss",d["config"]["final_loss"])
PY
done; done
