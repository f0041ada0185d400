#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "bf16x3" > gpurun_out/l_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/l_tests.log
tail -12 gpurun_out/l_tests.log
grep -q "rc 0" gpurun_out/l_tests.log || exit 1
for i in 1 2; do
for tm in 0 1; do
timeout -k 10 200 python bench.py --steps 240 --warmup 20 --no-cpu-baseline --no-extra-configs --no-gather-sweep --tower-mode $tm > gpurun_out/l_bench.json 2> gpurun_out/l_bench.err || { tail -5 gpurun_out/l_bench.err; exit 1; }
python - "$tm" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/l_bench.json").read().strip().splitlines()[-1])
print("tower_mode",sys.argv[1],"ms",round(d["ms_per_step"],4),"loss",d["config"]["final_loss"])
PY
done; done

