#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters per kernel: pmc_summary.py DIR [name-substring ...]"""
import collections, csv, glob, os, sys

root = sys.argv[1]
filt = sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if filt and not any(s in k for s in filt):
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add((f, r["Dispatch_Id"]))
for k in sorted(acc):
    print(f"{k[:90]}  ({len(disp[k])} dispatches)")
    for c in sorted(acc[k]):
        print(f"    {c:28s} {acc[k][c]:18.0f}")
