#!/usr/bin/env python3
"""Group a rocprofv3 --kernel-trace CSV by (kernel name, grid size): calls, avg / min / median us.
usage: tools/ktrace_groups.py <dir-or-kernel_trace.csv> [name-filter] [--runs]
--runs: one group per RUN of consecutive dispatches of the same (kernel, grid) in time order (the same
kernel timed in several contexts, separated by any other kernel, gives several lines)."""
import csv
import glob
import os
import statistics
import sys

runs = "--runs" in sys.argv
argv = [a for a in sys.argv if a != "--runs"]
path = argv[1]
flt = argv[2] if len(argv) > 2 else ""
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[0]
groups = {}
order = []
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
run_id, last = 0, None
for r in rows:
    name = r["Kernel_Name"]
    ident = (name, r.get("Grid_Size_X", r.get("Grid_Size", 0)))
    if ident != last:
        run_id, last = run_id + 1, ident
    if flt and flt not in name:
        continue
    grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
    wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0)
    key = (name, grid, wg, run_id if runs else 0)
    if key not in groups:
        groups[key] = []
        order.append(key)
    groups[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("kernel,grid,workgroup,calls,avg_us,median_us,min_us,max_us")
for key in order:
    v = groups[key]
    print(f"\"{key[0][:110]}\",{key[1]},{key[2]},{len(v)},{sum(v) / len(v):.2f},{statistics.median(v):.2f},{min(v):.2f},{max(v):.2f}")
