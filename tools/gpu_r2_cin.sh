#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/cin_layer -o run -- python3 $root/tools/time_layers.py cin 30 > $out/cin_layer.log 2>&1
tail -2 $out/cin_layer.log
python3 $root/tools/kstats.py $out/cin_layer 33 20
