#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for g in 128 300; do
for v in -1 0; do
  d=gpurun_out/prof_small_${g}_${v}
  rm -rf $d
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 scripts/small_grid_lab.py $v $g 200 hard > $d.log 2>&1
  echo "g=$g v=$v rc=$?"
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  head -12 "$f" | cut -c1-200
done
done
