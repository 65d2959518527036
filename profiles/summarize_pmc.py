#!/usr/bin/env python3
"""Summarises a profiles/collect_pmc.sh output directory: per-kernel mean duration
from the kernel trace and per-dispatch mean of every PMC counter."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("# kernel stats:", os.path.relpath(f, out))
    print(open(f).read())
vals = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "?")
        if "conv_" not in k and "concat" not in k:
            continue
        vals[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in vals.items():
    print("# PMC per dispatch (mean over %d dispatches):" % max(len(v) for v in cs.values()), k[:90])
    for c, v in sorted(cs.items()):
        print("%-32s %18.1f" % (c, sum(v) / len(v)))
