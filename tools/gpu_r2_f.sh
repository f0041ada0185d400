#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for sh in 4 5; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_sh$sh -o run -- python3 $root/bench.py --steps 200 --warmup 20 --no-cpu-baseline --gather-shape $sh > $out/bench_sh$sh.json 2> $out/bench_sh$sh.err || { tail -5 $out/bench_sh$sh.err; exit 1; }
  cat $out/bench_sh$sh.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shape', $sh, d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['min_launch_us'], d['config']['final_loss'])"
  python3 $root/tools/kstats.py $out/bench_sh$sh 220 30
done
