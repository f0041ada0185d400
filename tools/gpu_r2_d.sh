#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout -k 10 600 python3 -m pytest tests/test_gpu_fullsize.py -q -x > $out/r2d_pytest.log 2>&1
rc=$?
tail -12 $out/r2d_pytest.log
if [ $rc -ge 124 ]; then echo "pytest timed out"; exit $rc; fi
cd /tmp && export TMPDIR=/tmp
for mt in uncached finegrained; do
  export MB_TABLE_MEM=$mt
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/mb2d_$mt -o run -- $root/tools/microbench_gather2 4096 20 1 > $out/mb2d_$mt.log 2> $out/mb2d_$mt.err || { tail -5 $out/mb2d_$mt.err; exit 1; }
  python3 $root/tools/ktrace_groups.py $out/mb2d_$mt > $out/mb2d_${mt}_groups.csv
  echo "=== $mt"; grep -v rocclr $out/mb2d_${mt}_groups.csv | cut -c1-70,100-200
  for c in TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_32B_sum; do
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/mb2_pmc_$c -o run -- $root/tools/microbench_gather2 4096 4 > /dev/null 2> $out/mb2_pmc_$c.err || { echo "pmc $c failed"; tail -3 $out/mb2_pmc_$c.err; continue; }
    python3 $root/tools/pmc_summary.py $out/mb2_pmc_$c $c > $out/mb2d_${mt}_pmc_$c.txt
    rm -rf $out/mb2_pmc_$c
    echo "== $c"; grep -v rocclr $out/mb2d_${mt}_pmc_$c.txt | grep "probe_rows\|owner_pol<1, true, 0>\|emb_fwd"
  done
done
