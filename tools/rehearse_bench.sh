#!/bin/bash
# bench.py's N > 1 code path on a one-GPU box: every rank on cuda:0 over gloo (DFM_BENCH_REHEARSAL=1), both
# data-parallel layouts.  The timings mean nothing; the point is that the driver's command line runs.
# usage (GPU box): bash tools/rehearse_bench.sh [ranks=2]
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
n=${1:-2}
cd $root
export HSA_ENABLE_IPC_MODE_LEGACY=0 DFM_BENCH_REHEARSAL=1
for mode in sharded replicated; do
  timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29541 \
    bench.py --gpus $n --steps 8 --warmup 4 --vocab 100000 --dp-mode $mode > $out/rehearse_$mode.json 2> $out/rehearse_$mode.err
  rc=$?
  echo "$mode rc=$rc"
  if [ $rc -ne 0 ]; then grep -v "^  File\|^    " $out/rehearse_$mode.err | tail -12; exit 1; fi
  python3 -c "
import json
d=json.loads([l for l in open('$out/rehearse_$mode.json') if l.startswith('{')][-1])
print(d['n_gpus'], d['config']['parallelism'], 'loss', d['config']['final_loss'], 'lines on stdout:', sum(1 for _ in open('$out/rehearse_$mode.json')))"
done
