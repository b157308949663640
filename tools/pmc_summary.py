#!/usr/bin/env python3
"""summarise rocprofv3 --pmc csv output: mean counter value per kernel name (usage: pmc_summary.py <dir> [filter])"""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        if len(sys.argv) > 2 and sys.argv[2] not in r["Kernel_Name"]:
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:45s} mean={sum(v)/len(v):16.1f} n={len(v)}")
