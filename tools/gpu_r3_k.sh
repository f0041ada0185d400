#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "dnn or tower or gemm or layernorm or train_golden or models_step" > gpurun_out/k_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/k_tests.log
tail -5 gpurun_out/k_tests.log
for i in 1 2; do
for v in new old; do
if [ $v == new ]; then unset DFM_LIB_PATH; else export DFM_LIB_PATH=$PWD/deepfm_amd/lib/variants/lib_gemm_old.so; fi
timeout -k 10 200 python bench.py --steps 240 --warmup 20 --no-cpu-baseline --no-extra-configs --no-gather-sweep > gpurun_out/k_bench.json 2> gpurun_out/k_bench.err || { tail -5 gpurun_out/k_bench.err; exit 1; }
python - "$v" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/k_bench.json").read().strip().splitlines()[-1])
print(sys.argv[1],"ms",round(d["ms_per_step"],4),"loss",d["config"]["final_loss"])
PY
done; done
unset DFM_LIB_PATH
python tools/time_linear_bwd.py 2>&1 | tail -4
