#!/bin/bash
# GPU box: rocprofv3 kernel stats + one PMC pass of the value+grad sweep with the NaN scan in both shapes.
#   usage: prof_nan_scan.sh [tag] [cases...]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${1:-r04_nan}; shift || true
CASES=${*:-"cfg3 cfg3_hsig"}
OUT=gpurun_out/${TAG}
mkdir -p $OUT
for c in $CASES; do
  for m in 1 2; do
    export D2D_NAN_SCAN=$m
    rm -rf $OUT/${c}_m${m}_trace $OUT/${c}_m${m}_pmc
    timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${c}_m${m}_trace -- python3 scripts/kernel_lab.py $c 20 > $OUT/${c}_m${m}_trace.log 2>&1
    echo "$c m$m trace rc=$?"
    timeout -k 10 150 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/${c}_m${m}_pmc -- python3 scripts/kernel_lab.py $c 6 > $OUT/${c}_m${m}_pmc.log 2>&1
    echo "$c m$m pmc rc=$?"
    f=$(ls $OUT/${c}_m${m}_trace/*/*kernel_stats.csv 2>/dev/null | head -1)
    [ -n "$f" ] && head -8 "$f" | cut -c1-200
  done
done
