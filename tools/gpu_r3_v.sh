#!/bin/bash
# rehearsals of the N > 1 bench path on the one-GPU box: two ranks over gloo on cuda:0, then one rank over RCCL with the
# data-parallel structure forced
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
DFM_BENCH_REHEARSAL=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/v_gloo2.json 2> gpurun_out/v_gloo2.err
echo "gloo2 rc $?"; tail -c 900 gpurun_out/v_gloo2.json; echo
DFM_FORCE_DP_PATH=1 timeout -k 10 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-extra-configs --no-gather-sweep > gpurun_out/v_rccl1.json 2> gpurun_out/v_rccl1.err
echo "rccl1 rc $?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/v_rccl1.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["config"].get("hip_graph"), d["config"].get("capture_fallback"), d["config"].get("dp_layout"))
PY
