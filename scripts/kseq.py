#!/usr/bin/env python3
"""The last N dispatches of a rocprofv3 kernel trace in start order: start offset, duration, gap to the previous end (us), stream
queue and name.   usage: kseq.py <trace dir> [N]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-n:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = "" if prev_end is None else f"{(s - prev_end) / 1e3:8.1f}"
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:>8s}  q{r.get('Queue_Id', '?'):>3s}  {r['Kernel_Name'][:90]}")
    prev_end = max(e, prev_end or e)
