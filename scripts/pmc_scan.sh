#!/bin/bash
# GPU box: PMC counters of the NaN scan kernel at cfg3 (two rocprofv3 --pmc passes, each on its own), printed per dispatch.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-scan}_pmc
mkdir -p $OUT
for p in 1 2; do
  if [ $p = 1 ]; then C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; else C="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"; fi
  rm -rf $OUT/p$p
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$p -- python3 scripts/kernel_lab.py ${2:-cfg3} 10 > $OUT/p$p.log 2>&1
  echo "pass $p rc=$?"
done
python3 - <<PY
import csv, glob, collections
for p in (1, 2):
    files = glob.glob("$OUT/p%d/**/*counter_collection.csv" % p, recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "nan_scan" in k or "power_fwd_kernel" in k:
                acc[k[:60]][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k[:60], r["Counter_Name"])] += 1
    for k, d in acc.items():
        print(k)
        for c, v in d.items(): print("   %-24s %.4g per dispatch (%d)" % (c, v / n[(k, c)], n[(k, c)]))
PY
