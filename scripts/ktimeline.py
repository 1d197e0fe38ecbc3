#!/usr/bin/env python3
"""Start / end (us, relative) of every dispatch in a window of a rocprofv3 kernel trace: do kernels of different streams overlap?"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 30
t0 = int(rows[skip]["Start_Timestamp"])
for r in rows[skip : skip + n]:
    a, b = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{a:9.1f} {b:9.1f}  q{r.get('Queue_Id', '?'):>3s}  {r['Kernel_Name'][:70]}")
