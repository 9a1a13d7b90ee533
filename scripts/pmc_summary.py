#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc CSVs (scripts/pmc.sh output) per kernel: mean of each counter per dispatch."""
import csv, glob, sys, collections, json
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{root}/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:90]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in acc.items():
    out[k] = {c: sum(v) / len(v) for c, v in d.items()}
    out[k]["dispatches"] = max(len(v) for v in d.values())
json.dump(out, open(f"{root}/summary.json", "w"), indent=1)
for k, d in sorted(out.items(), key=lambda x: -x[1].get("SQ_WAVE_CYCLES", 0))[:4]:
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {v:16.1f}")
