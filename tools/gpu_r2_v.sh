#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
cd $root
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== plain N=1"
python3 bench.py --steps 200 --warmup 40 --no-cpu-baseline --no-extra-configs > $out/r2v_plain.json 2> $out/r2v_plain.err || { tail -5 $out/r2v_plain.err; exit 1; }
python3 -c "import json;d=json.load(open('$out/r2v_plain.json'));print(d['ms_per_step'], d['value'], d['config']['final_loss'])"
echo "== sharded structure, one rank over RCCL (in-graph collectives)"
DFM_FORCE_DP_PATH=1 timeout -k 10 300 python3 bench.py --steps 200 --warmup 40 --no-cpu-baseline --no-extra-configs > $out/r2v_sharded.json 2> $out/r2v_sharded.err || { tail -5 $out/r2v_sharded.err; exit 1; }
python3 -c "import json;d=json.load(open('$out/r2v_sharded.json'));print(d['ms_per_step'], d['value'], d['config']['final_loss'], d['config']['parallelism'])"
echo "== replicated structure, one rank over RCCL (split path)"
DFM_FORCE_DP_PATH=1 timeout -k 10 300 python3 bench.py --steps 200 --warmup 40 --no-cpu-baseline --no-extra-configs --dp-mode replicated > $out/r2v_repl.json 2> $out/r2v_repl.err || { tail -5 $out/r2v_repl.err; exit 1; }
python3 -c "import json;d=json.load(open('$out/r2v_repl.json'));print(d['ms_per_step'], d['value'], d['config']['final_loss'], d['config']['parallelism'])"
echo "== rehearsal: 2 ranks gloo on cuda:0, sharded"
DFM_BENCH_REHEARSAL=1 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 6 --warmup 2 --vocab 100000 > $out/r2v_reh.json 2> $out/r2v_reh.err || { tail -5 $out/r2v_reh.err; exit 1; }
tail -c 600 $out/r2v_reh.json
