#!/bin/bash
# timing-only ablation of rowplan_sort (run on the GPU box): 0 full, 1 no sort, 2 no output phase, 3 neither
cd /tmp && export TMPDIR=/tmp
for a in 0 1 2 3; do
  DFM_ROWPLAN_ABLATE=$a rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp$a -o run -- python3 $GRAFT_REPO_ROOT/tools/rp_time.py > /dev/null 2>&1
  python3 - <<PY
import csv
for r in csv.DictReader(open("/tmp/rp$a/run_kernel_stats.csv")):
    if "rowplan_sort" in r["Name"]:
        print("ABLATE $a avg_us", float(r["AverageNs"]) / 1e3, "min", float(r["MinNs"]) / 1e3)
PY
done
