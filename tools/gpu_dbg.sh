#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
cd $root
export HSA_ENABLE_IPC_MODE_LEGACY=0
for kind in sharded fused; do
timeout -k 5 200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29517 tests/dp_rehearsal_worker.py eager 3 $kind > $out/dbg_$kind.log 2>&1
echo "$kind rc=$?"
grep -n "Error\|error" $out/dbg_$kind.log | grep -v "Connection reset\|ChildFailed\|error_file" | head -8
done
