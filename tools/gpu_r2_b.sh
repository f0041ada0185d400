#!/bin/bash
# round 2, GPU call B: full-size test again + microbench v2 (policies, stamps) + PMC passes on the probes
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout -k 10 600 python3 -m pytest tests/test_gpu_fullsize.py -q -x > $out/r2b_pytest.log 2>&1
rc=$?
tail -15 $out/r2b_pytest.log
if [ $rc -ge 124 ]; then echo "pytest timed out"; exit $rc; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/mb2b -o run -- $root/tools/microbench_gather2 4096 30 > $out/mb2b.log 2> $out/mb2b.err || { tail -5 $out/mb2b.err; exit 1; }
python3 $root/tools/ktrace_groups.py $out/mb2b > $out/mb2b_groups.csv
cat $out/mb2b.log
rocprofv3 -L > $out/avail.txt 2>&1
grep -o "TCC_EA[A-Z0-9_]*\|TCP_[A-Z0-9_]*TCC[A-Z0-9_]*" $out/avail.txt | sort -u | head -80 > $out/avail_tcc.txt
for c in FETCH_SIZE TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_sum WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/mb2_pmc_$c -o run -- $root/tools/microbench_gather2 4096 4 > /dev/null 2> $out/mb2_pmc_$c.err || { echo "pmc $c failed"; tail -3 $out/mb2_pmc_$c.err; continue; }
  python3 $root/tools/pmc_summary.py $out/mb2_pmc_$c $c > $out/mb2_pmc_$c.txt
  rm -rf $out/mb2_pmc_$c
  echo "== $c"; cat $out/mb2_pmc_$c.txt
done
