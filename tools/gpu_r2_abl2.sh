#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
cd $root
timeout -k 10 300 python3 -m pytest tests/test_gpu_cin.py -q -x 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for a in 0 7; do
DFM_WG_ABL=$a timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/abl$a -o run -- python3 $root/tools/time_layers.py cin 12 > $out/abl$a.log 2>&1
echo "ABL=$a"; python3 $root/tools/kstats.py $out/abl$a 15 4 | grep wgrad_mfma
done
