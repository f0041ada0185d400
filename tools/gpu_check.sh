#!/bin/bash
# usage: tools/gpu_check.sh <pytest args...>   (GPU box: run the given tests, print the tail / the failures)
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
cd $root
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 1000 python3 -m pytest "$@" -q -x -p no:cacheprovider > $out/check_pytest.log 2>&1
rc=$?
tail -4 $out/check_pytest.log
if [ $rc -ne 0 ]; then grep -E "^E |^FAILED" $out/check_pytest.log | head -30; exit 1; fi
