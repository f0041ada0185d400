#!/usr/bin/env python3
"""Group a rocprofv3 --kernel-trace CSV by (kernel name, grid size): calls, avg / min / median us.
usage: tools/ktrace_groups.py <dir-or-kernel_trace.csv> [name-filter]"""
import csv
import glob
import os
import statistics
import sys

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[0]
groups = {}
order = []
for r in csv.DictReader(open(path)):
    name = r["Kernel_Name"]
    if flt and flt not in name:
        continue
    grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
    wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0)
    key = (name, grid, wg)
    if key not in groups:
        groups[key] = []
        order.append(key)
    groups[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("kernel,grid,workgroup,calls,avg_us,median_us,min_us,max_us")
for key in order:
    v = groups[key]
    print(f"\"{key[0][:110]}\",{key[1]},{key[2]},{len(v)},{sum(v) / len(v):.2f},{statistics.median(v):.2f},{min(v):.2f},{max(v):.2f}")
