#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel name and counter, mean per dispatch."""
import csv, sys, collections, glob
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0]
            if "ldpc" not in k and "tb_" not in k: continue
            acc[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print(f"{f.split('/')[-3]:10s} {k:28s} {c:26s} n={len(v):3d} mean={sum(v)/len(v):16.1f} max={max(v):16.1f}")
