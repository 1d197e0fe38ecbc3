#!/bin/bash
# A/B helper (GPU box): builds libd2d variants with extra -D flags ASIDE (under /tmp, selected through the D2D_LIB
# environment variable -- the product library differt2d_amd/csrc/libd2d.so is never touched) and benches each.
#   AB_ARCH overrides the offload target (e.g. gfx950:xnack-)
#   usage: ab_build.sh "tag1:-DX=1" "tag2:-DY=2" ...      (AB_CMD overrides the command run against each variant)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for spec in "$@"; do
  tag="${spec%%:*}"; flags="${spec#*:}"
  lib="/tmp/d2d_ab_${tag}/libd2d.so"
  make -C differt2d_amd/csrc -s -j 16 ARCH="${AB_ARCH:-gfx950}" B="/tmp/d2d_ab_${tag}/build" OUT="$lib" EXTRA="$flags" > gpurun_out/ab_${tag}_build.log 2>&1 || { echo "$tag build failed"; continue; }
  if [ -n "${AB_CMD:-}" ]; then
    echo "== $tag"; D2D_LIB="$lib" bash -c "$AB_CMD"
    continue
  fi
  for approx in 0 1; do
    D2D_LIB="$lib" python bench.py --steps 50 --no-cpu-baseline ${AB_BENCH_FLAGS:---no-extras} --approx $approx > gpurun_out/ab_${tag}_a${approx}.log 2>&1
    python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/ab_${tag}_a${approx}.log").read().strip().splitlines()[-1])
    print("${tag}", "approx=${approx}", "ms=%.3f"%d["ms_per_step"], "kernel_ms=%.3f"%d["roofline"]["kernel_ms"])
except Exception as e:
    print("${tag}", "approx=${approx}", "FAILED", e)
PY
  done
done
