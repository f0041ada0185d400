#!/bin/bash
# round 2, GPU call A: new parity tests + the gather microbenchmark ladder under rocprofv3
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout -k 10 900 python3 -m pytest tests/test_gpu_train_golden.py tests/test_gpu_fullsize.py tests/test_gpu_dnn_fused.py \
    tests/test_gpu_models_step.py tests/test_gpu_fused_tower.py tests/test_gpu_auc_parity.py -q > $out/r2a_pytest.log 2>&1
rc=$?
tail -5 $out/r2a_pytest.log
if [ $rc -ge 124 ]; then echo "pytest timed out"; exit $rc; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/mb2 -o run -- $root/tools/microbench_gather2 4096 30 > $out/mb2.log 2> $out/mb2.err || { tail -5 $out/mb2.err; exit 1; }
python3 $root/tools/ktrace_groups.py $out/mb2 > $out/mb2_groups.csv
cat $out/mb2.log
