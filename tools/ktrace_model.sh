#!/bin/bash
# Kernel statistics of one whole-model training step (GPU box): rocprofv3 kernel trace of tools/time_models.py.
# usage: bash tools/ktrace_model.sh <deepfm|xdeepfm|attention_deepfm> [steps]
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
model=${1:-attention_deepfm}
steps=${2:-40}
cd /tmp && export TMPDIR=/tmp
rm -rf $out/ktm
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ktm -o run -- python3 $root/tools/time_models.py $steps $model > $out/ktm.log 2>&1 || { tail -5 $out/ktm.log; exit 1; }
tail -2 $out/ktm.log
python3 $root/tools/kstats.py $out/ktm $((steps + 12)) 34 | cut -c1-60,97-200
