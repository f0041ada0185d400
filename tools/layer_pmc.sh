#!/bin/bash
# One layer (cin | attn at the BASELINE.json configuration 3 / 4 shapes): kernel trace + one PMC pass per counter
# (rocprofv3 --pmc alone with --kernel-trace).  usage (GPU box): bash tools/layer_pmc.sh <cin|attn>
#   -> gpurun_out/<layer>_pmc/{kernel_stats.csv,pmc.txt}
layer=${1:-cin}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/${layer}_pmc
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o run -- python3 $root/tools/time_layers.py $layer 12 > $out/time.txt 2> $out/trace.err || exit 1
cp $(ls $out/trace/*kernel_stats.csv | head -1) $out/kernel_stats.csv
: > $out/pmc.txt
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -o run -- python3 $root/tools/time_layers.py $layer 4 > /dev/null 2> $out/pmc_$c.err || exit 1
  python3 $root/tools/pmc_summary.py $out/pmc_$c $c $layer >> $out/pmc.txt
  rm -rf $out/pmc_$c
  echo "done $c"
done
