#!/usr/bin/env python3
"""Average a rocprofv3 --pmc counter per kernel. usage: tools/pmc_summary.py <dir> <COUNTER> [name-substring]
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
(MI355X_MICROARCH.md, HBM section) — the caller applies that correction."""
import csv
import glob
import os
import sys
from collections import defaultdict

d, counter = sys.argv[1], sys.argv[2]
sub = sys.argv[3] if len(sys.argv) > 3 else ""
path = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))[0]
tot, cnt = defaultdict(float), defaultdict(int)
for r in csv.DictReader(open(path)):
    if r["Counter_Name"] != counter:
        continue
    name = r["Kernel_Name"][:70]
    if sub and sub not in name:
        continue
    tot[name] += float(r["Counter_Value"])
    cnt[name] += 1
for name in sorted(tot, key=lambda k: -tot[k])[:25]:
    print(f"{name:70s} n={cnt[name]:5d} avg {counter}={tot[name] / cnt[name]:12.2f}")
