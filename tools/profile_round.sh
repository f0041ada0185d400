#!/bin/bash
# Runs on the GPU box (gpurun): headline bench + rocprofv3 kernel trace + separate PMC passes for the
# embedding gather.  Everything lands under gpurun_out/<tag>/; copy what is judged into profiles/.
# usage: tools/profile_round.sh <tag>
set -o pipefail
tag=${1:-r01}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $root/bench.py --steps 300 --warmup 20 > $out/bench.json 2> $out/bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o run -- python3 $root/bench.py --steps 100 --warmup 20 --no-cpu-baseline > $out/bench_traced.json 2> $out/trace.err || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -o run -- python3 $root/bench.py --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2> $out/pmc_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -o run -- python3 $root/bench.py --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2> $out/pmc_write.err || exit 1
python3 $root/tools/pmc_summary.py $out/pmc_fetch FETCH_SIZE emb_fwd_uniform > $out/pmc_fetch.txt
python3 $root/tools/pmc_summary.py $out/pmc_write WRITE_SIZE emb_fwd_uniform > $out/pmc_write.txt
cat $out/bench.json $out/pmc_fetch.txt $out/pmc_write.txt
