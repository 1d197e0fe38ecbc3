#!/usr/bin/env python3
"""Prints the per-kernel averages of a rocprofv3 --kernel-trace --stats output directory."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    print(f"{r['Name'][:80]:80s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.2f} us")
