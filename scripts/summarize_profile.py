#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>_a<approx>_* (rocprofv3 CSV output of scripts/profile_gpu.sh)
into profiles/<tag>_a<approx>_{kernel_stats.csv,pmc.json,summary.md}."""

import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)

def is_main(name):
    """The timed sweep kernel: power_fwd_kernel<MODE, STATS = false, MAXK, GRADK = false, LISTED = true, WPB> (the instrumented
    build has STATS = true; LISTED = false is the enumerating build that walks the queue of left-over patches)."""
    if "power_fwd_kernel<" not in name:
        return False
    args = [x.strip() for x in name.split("<", 1)[1].split(">")[0].split(",")]
    return len(args) >= 5 and args[1] == "false" and args[3] == "false" and args[4] == "true"


for approx in (0, 1):
    base = os.path.join(root, "gpurun_out", f"prof_{tag}_a{approx}")
    # gpurun merges every call's output into gpurun_out/: keep the newest run of each pass only
    stats = sorted(glob.glob(base + "_trace/*/*_kernel_stats.csv"), key=os.path.getmtime, reverse=True)
    if not stats:
        continue
    dst = os.path.join(root, "profiles", f"{tag}_a{approx}_kernel_stats.csv")
    shutil.copy(stats[0], dst)
    pmc = {}
    newest = {}
    for f in glob.glob(base + "_pmc*/*/*_counter_collection.csv"):
        d = os.path.dirname(f)
        if d not in newest or os.path.getmtime(f) > os.path.getmtime(newest[d]):
            newest[d] = f
    for f in newest.values():
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if is_main(r["Kernel_Name"]):
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                pmc.setdefault("_meta", {"VGPR": r["VGPR_Count"], "SGPR": r["SGPR_Count"], "grid": r["Grid_Size"],
                                          "workgroup": r["Workgroup_Size"], "kernel": r["Kernel_Name"]})
        for k, v in agg.items():
            pmc[k] = {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)}
    json.dump(pmc, open(os.path.join(root, "profiles", f"{tag}_a{approx}_pmc.json"), "w"), indent=1)
    rows = list(csv.DictReader(open(stats[0])))
    k = max((r for r in rows if is_main(r["Name"])), key=lambda r: float(r["TotalDurationNs"]))
    avg_ms = float(k["AverageNs"]) / 1e6
    g = lambda n: pmc.get(n, {}).get("mean_per_dispatch", float("nan"))
    cycles_per_xcd = g("GRBM_GUI_ACTIVE") / 8
    simd_cycles = cycles_per_xcd * 1024
    with open(os.path.join(root, "profiles", f"{tag}_a{approx}_summary.md"), "w") as f:
        f.write(f"# rocprofv3 summary, tag {tag}, approx={approx}\n\n")
        f.write(f"command: `python3 bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-extras --approx {approx}` (scripts/profile_gpu.sh)\n\n")
        f.write(f"| kernel | calls | avg ms | min ms | max ms |\n|---|---|---|---|---|\n")
        f.write(f"| `{k['Name']}` | {k['Calls']} | {avg_ms:.4f} | {float(k['MinNs'])/1e6:.4f} | {float(k['MaxNs'])/1e6:.4f} |\n\n")
        f.write("All kernels of the run (one launch sequence per step: memset, shadow masks, region lists, schedule sort on a side "
                "stream, sweep, queue of left-over patches):\n\n| kernel | calls | avg us |\n|---|---|---|\n")
        for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
            f.write(f"| `{r['Name'][:110]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} |\n")
        f.write("\nPMC of the sweep kernel (mean per dispatch, separate passes):\n\n| counter | value |\n|---|---|\n")
        for name in sorted(n for n in pmc if n != "_meta"):
            f.write(f"| {name} | {g(name):.6g} |\n")
        f.write(f"\nDerived: clock ~ {cycles_per_xcd / (avg_ms * 1e-3) / 1e9:.2f} GHz (GRBM_GUI_ACTIVE/8/time); "
                f"VALU wave-instructions per wave {g('SQ_INSTS_VALU') / g('SQ_WAVES'):.0f}; "
                f"VALU issue utilisation at 2 cycles per wave64 instruction on 1024 SIMDs = "
                f"{2 * g('SQ_INSTS_VALU') / simd_cycles:.3f}; "
                f"executed lane-ops/s = {64 * g('SQ_INSTS_VALU') / (avg_ms * 1e-3) / 1e12:.1f} T/s "
                f"(non-FMA peak 78.6 T/s, FMA peak 157.3 TFLOP/s).\n"
                f"HBM: FETCH_SIZE {g('FETCH_SIZE'):.0f} KiB = {g('FETCH_SIZE') * 1024 / 1e6:.1f} MB as counted (these are "
                f"4-byte-per-lane loads: MI355X_MICROARCH.md's x2 correction is calibrated for 16-B-per-lane streaming reads "
                f"only, so {2 * g('FETCH_SIZE') * 1024 / 1e6:.1f} MB is the upper bound), WRITE_SIZE {g('WRITE_SIZE'):.0f} KiB "
                f"({g('WRITE_SIZE') * 1024 / 1e6:.1f} MB) per launch; algorithmic 12.6 MB (8.4 read + 4.2 written). "
                f"bench.py's roofline.traffic = FETCH_SIZE + WRITE_SIZE as counted.\n")
    print(open(os.path.join(root, "profiles", f"{tag}_a{approx}_summary.md")).read())
