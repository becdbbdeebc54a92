#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter rows per kernel over the passes found under a directory."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row.get("Kernel_Name") or row.get("Kernel_name") or "?"
            k = k.replace("(anonymous namespace)::", "").split("(")[0]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in sorted(acc.items()):
    n = max(len(v) for v in cs.values())
    print(f"== {k}  ({n} dispatches)")
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} avg {sum(v) / len(v):16.1f}   min {min(v):16.1f} max {max(v):16.1f}")
