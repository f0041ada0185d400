#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
cd $root
export HSA_ENABLE_IPC_MODE_LEGACY=0
for mode in eager graph; do
  echo "== $mode"
  DFM_WORKER_TRACE=1 timeout -k 5 150 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
    tests/dp_rehearsal_worker.py $mode 4 sharded nccl > $out/r2u_$mode.log 2>&1
  echo "rc=$?"
  tail -5 $out/r2u_$mode.log
done
