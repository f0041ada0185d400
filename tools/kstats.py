#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats CSV: per-kernel launches/step and us/step.
usage: tools/kstats.py <dir-or-kernel_stats.csv> [steps] [top]"""
import csv
import glob
import os
import sys

path = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(path)))
for r in rows[:top]:
    print(f"{r['Name'][:96]:96s} {int(r['Calls']) / steps:7.1f}/step avg_us={float(r['AverageNs']) / 1e3:8.2f} "
          f"min={float(r['MinNs']) / 1e3:7.2f} us/step={float(r['TotalDurationNs']) / 1e3 / steps:8.1f}")
print("launches/step", round(sum(int(r["Calls"]) for r in rows) / steps, 1),
      " gpu us/step", round(sum(float(r["TotalDurationNs"]) for r in rows) / 1e3 / steps, 1))
