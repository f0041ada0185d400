#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
DFM_BENCH_REHEARSAL=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 3 --steps 20 --warmup 5 > gpurun_out/z_gloo3.json 2> gpurun_out/z_gloo3.err
echo "gloo3 sharded rc $?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/z_gloo3.json").read().strip().splitlines()[-1])
print(d["n_gpus"], d["ms_per_step"], d["config"]["parallelism"], d["config"].get("hip_graph"), d["config"].get("capture_fallback"), d["config"]["final_loss"])
PY
DFM_BENCH_REHEARSAL=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 2 --steps 20 --warmup 5 --dp-mode replicated > gpurun_out/z_gloo2r.json 2> gpurun_out/z_gloo2r.err
echo "gloo2 replicated rc $?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/z_gloo2r.json").read().strip().splitlines()[-1])
print(d["n_gpus"], d["ms_per_step"], d["config"]["parallelism"], d["config"].get("hip_graph"), d["config"].get("capture_fallback"), d["config"]["final_loss"])
PY
