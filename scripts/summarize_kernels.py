#!/usr/bin/env python3
"""Condenses gpurun_out/<tag>_kernels/ (scripts/profile_kernels.sh: rocprofv3 --kernel-trace --stats and separate --pmc passes of
every scripts/kernel_lab.py case) into profiles/<tag>_kernels_summary.md, profiles/<tag>_kernels_pmc.json and one
profiles/<tag>_<case>_kernel_stats.csv per case."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"{tag}_kernels")
MAIN = {"cfg3": "power_fwd_kernel<0, false, 2, true", "cfg3_hsig": "power_fwd_kernel<1, false, 2, true", "txg": "power_fwd_txg_kernel",
        "cfg4": "power_fwd_kernel<0, false, 3, false, true", "sigmoid": "power_fwd_kernel<2, false, 2, false, true",
        "cfg5": "power_opt_rev_kernel", "cfg5_fwd": "power_opt_cand_kernel", "cfg5_tan": "power_opt_grad_kernel"}
WHAT = {"cfg3": "cfg2's sweep with value + per-cell gradient + scene VJP, hard (BASELINE.json configs[2])", "cfg3_hsig": "the same, hard_sigmoid",
        "txg": "TX grid of cfg2's size, hard (accumulate_on_transmitters_grid_over_paths)", "cfg4": "200 walls, 2048^2, orders 0..3, hard (configs[3])",
        "sigmoid": "cfg2 in sigmoid validity", "cfg5": "RIS scene, 300^2, MinPath x 1000 steps, value + gradient + scene VJP, reverse mode (configs[4])",
        "cfg5_fwd": "the same, values only", "cfg5_tan": "the same gradients by forward tangents (opt_grad_mode 1: round 2's kernel)"}
# (second kernels of a case's run: the NaN scan beside the value+grad sweep -- VERDICT r5 item 4 asks for its counters)
SECOND = {"cfg3_scan": ("cfg3", "nan_scan_region_kernel"), "cfg3_hsig_scan": ("cfg3_hsig", "nan_scan_region_kernel")}
for k_, (dir_, key_) in SECOND.items():
    MAIN[k_] = key_
    WHAT[k_] = f"the NaN scan that runs beside the sweep of case {dir_} (d2d_nanscan.hpp)"
PEAK = 157.3e12
out_md = [f"# rocprofv3 summaries of the non-headline kernels, tag {tag}\n",
          "`scripts/profile_kernels.sh` -> `scripts/kernel_lab.py <case>` under `rocprofv3 --kernel-trace --stats` and, each on its own, the `--pmc` "
          "passes of `scripts/profile_gpu.sh`; condensed by `scripts/summarize_kernels.py`.  `lane-ops frac` = 64 x SQ_INSTS_VALU / kernel time / "
          "157.3 T/s (an executed-instruction rate: FMA counted once, masked lanes counted); `VALU util` = 2 cycles per wave64 instruction over "
          "1024 SIMDs; `wait` = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES; HBM = FETCH_SIZE / WRITE_SIZE as counted (KiB -> MB).\n",
          "| case | kernel | calls | avg ms | VGPR | AGPR | SGPR | scratch B | LDS B | waves | VALU instr / launch | lane-ops frac | VALU util | wait | HBM rd MB | HBM wr MB |",
          "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
allpmc = {}
for case, key in MAIN.items():
    rundir = SECOND[case][0] if case in SECOND else case
    stats = sorted(glob.glob(os.path.join(src, f"{rundir}_trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime, reverse=True)
    if not stats:
        continue
    if case not in SECOND:
        shutil.copy(stats[0], os.path.join(root, "profiles", f"{tag}_{case}_kernel_stats.csv"))
    rows = [r for r in csv.DictReader(open(stats[0])) if key in r["Name"]]
    if not rows:
        continue
    k = max(rows, key=lambda r: float(r["TotalDurationNs"]))
    pmc, meta = {}, {}
    for f in glob.glob(os.path.join(src, f"{rundir}_pmc*", "*", "*_counter_collection.csv")):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            if key in r["Kernel_Name"]:
                agg[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
                meta = {"VGPR": r["VGPR_Count"], "AGPR": r["Accum_VGPR_Count"], "SGPR": r["SGPR_Count"], "scratch": r["Scratch_Size"],
                        "LDS": r["LDS_Block_Size"], "grid": r["Grid_Size"], "workgroup": r["Workgroup_Size"], "kernel": r["Kernel_Name"]}
        for c, per in agg.items():
            v = list(per.values())
            pmc[c] = sum(v) / len(v)
    allpmc[case] = {"what": WHAT[case], "kernel_avg_ms": float(k["AverageNs"]) / 1e6, "calls": int(k["Calls"]), "meta": meta, "pmc_mean_per_dispatch": pmc}
    g = lambda n: pmc.get(n, float("nan"))
    t = float(k["AverageNs"]) * 1e-9
    simd_cycles = g("GRBM_GUI_ACTIVE") / 8 * 1024
    out_md.append(f"| {case} | `{k['Name'][:70]}` | {k['Calls']} | {t * 1e3:.4f} | {meta.get('VGPR')} | {meta.get('AGPR')} | {meta.get('SGPR')} | "
                  f"{meta.get('scratch')} | {meta.get('LDS')} | {g('SQ_WAVES'):.0f} | {g('SQ_INSTS_VALU'):.4g} | {64 * g('SQ_INSTS_VALU') / t / PEAK:.3f} | "
                  f"{2 * g('SQ_INSTS_VALU') / simd_cycles:.3f} | {g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'):.2f} | "
                  f"{g('FETCH_SIZE') * 1024 / 1e6:.1f} | {g('WRITE_SIZE') * 1024 / 1e6:.1f} |")
out_md.append("\nCases:\n")
for case in allpmc:
    out_md.append(f"* **{case}** -- {WHAT[case]}")
json.dump(allpmc, open(os.path.join(root, "profiles", f"{tag}_kernels_pmc.json"), "w"), indent=1)
open(os.path.join(root, "profiles", f"{tag}_kernels_summary.md"), "w").write("\n".join(out_md) + "\n")
print("\n".join(out_md))
