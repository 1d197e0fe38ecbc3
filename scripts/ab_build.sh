#!/bin/bash
# A/B helper (GPU box): builds libd2d variants with extra -D flags and benches each.  usage: ab_build.sh "tag1:-DX=1" "tag2:-DY=2" ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
cp differt2d_amd/csrc/libd2d.so /tmp/libd2d_orig.so
for spec in "$@"; do
  tag="${spec%%:*}"; flags="${spec#*:}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt $flags -o differt2d_amd/csrc/libd2d.so differt2d_amd/csrc/d2d.hip 2> gpurun_out/ab_${tag}_build.log || { echo "$tag build failed"; continue; }
  for approx in 0 1; do
    python bench.py --steps 20 --no-cpu-baseline ${AB_BENCH_FLAGS:---no-grad} --approx $approx > gpurun_out/ab_${tag}_a${approx}.log 2>&1
    python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/ab_${tag}_a${approx}.log").read().strip().splitlines()[-1])
    print("${tag}", "approx=${approx}", "ms=%.3f"%d["ms_per_step"], "kernel_ms=%.3f"%d["roofline"]["kernel_ms"], "vg_ms=%.3f"%d.get("value_and_grad",{}).get("ms_per_step",float("nan")))
except Exception as e:
    print("${tag}", "approx=${approx}", "FAILED", e)
PY
  done
done
cp /tmp/libd2d_orig.so differt2d_amd/csrc/libd2d.so
