#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout -k 10 1000 python3 -m pytest tests -m gpu -q -x > $out/r2e_pytest.log 2>&1
rc=$?
tail -12 $out/r2e_pytest.log
if [ $rc -ge 124 ]; then echo "pytest timed out"; exit $rc; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/tg -o run -- python3 $root/tools/time_gather.py 30 > $out/tg.log 2> $out/tg.err || { tail -5 $out/tg.err; exit 1; }
python3 $root/tools/ktrace_groups.py $out/tg emb_fwd_uniform --runs > $out/tg_groups.csv
cut -c1-60,160-260 $out/tg_groups.csv
