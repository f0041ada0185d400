#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
echo "== single-rank RCCL, split path"
DFM_FORCE_DP_PATH=1 timeout -k 10 300 python3 $root/bench.py --steps 100 --no-cpu-baseline --no-extra-configs > $out/bench_o1.json 2> $out/bench_o1.err || { tail -20 $out/bench_o1.err; exit 1; }
python3 -c "
import json
d=json.loads(open('$out/bench_o1.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['final_loss'], d['config']['env_switches'])"
echo "== single-rank RCCL, exchange inside the graph"
DFM_FORCE_DP_PATH=1 DFM_DP_GRAPH_COLLECTIVE=1 timeout -k 10 300 python3 $root/bench.py --steps 100 --no-cpu-baseline --no-extra-configs > $out/bench_o2.json 2> $out/bench_o2.err || { tail -20 $out/bench_o2.err; exit 1; }
python3 -c "
import json
d=json.loads(open('$out/bench_o2.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['final_loss'])"
echo "== 2 ranks on one GPU over gloo (rehearsal)"
DFM_BENCH_REHEARSAL=1 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 $root/bench.py --gpus 2 --steps 40 --warmup 8 --vocab 200000 > $out/bench_o3.json 2> $out/bench_o3.err || { tail -20 $out/bench_o3.err; exit 1; }
tail -1 $out/bench_o3.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['n_gpus'], d['config']['final_loss'])"
