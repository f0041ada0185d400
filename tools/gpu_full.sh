#!/bin/bash
# Full GPU check of a round: the -m gpu suite, smoke(), the default bench line.
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
cd $root
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 1000 python3 -m pytest tests -q -x -m gpu -p no:cacheprovider > $out/full_pytest.log 2>&1 &
pid=$!
while kill -0 $pid 2>/dev/null; do sleep 45; echo "pytest running: $(tail -c 120 $out/full_pytest.log | tr '\n' ' ')"; done
wait $pid; rc=$?
tail -4 $out/full_pytest.log
if [ $rc -ne 0 ]; then grep -E "^E |^FAILED" $out/full_pytest.log | head -30; exit 1; fi
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
