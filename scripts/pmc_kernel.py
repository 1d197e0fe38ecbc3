#!/usr/bin/env python3
"""Mean per dispatch of every counter in a rocprofv3 --pmc output directory, for kernels whose name contains argv[2]."""
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, c), v in per.items():
        acc[c].append(v)
for c, v in sorted(acc.items()):
    print(f"{c:28s} {sum(v)/len(v):.4g}  ({len(v)} dispatches)")
