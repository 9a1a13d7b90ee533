#!/usr/bin/env python3
"""Top kernels of a rocprofv3 --kernel-trace --stats output directory: scripts/kernel_stats_top.py <dir> [n]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for r in list(csv.DictReader(open(f)))[:n]:
    print("%9.1f us x%5s %6s%%  %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], r["Percentage"], r["Name"][:90]))
