#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/attn_step -o run -- python3 $root/tools/time_models.py 50 attention_deepfm > $out/attn_step.log 2>&1
python3 $root/tools/kstats.py $out/attn_step 60 40
