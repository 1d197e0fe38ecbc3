#!/bin/bash
# A/B helper (GPU box): builds libd2d variants with extra -D flags and runs scripts/sweep_bench.sh on each.  usage: ab_sweep.sh "tag:-DX=1" ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp differt2d_amd/csrc/libd2d.so /tmp/libd2d_orig.so
for spec in "$@"; do
  tag="${spec%%:*}"; flags="${spec#*:}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -DD2D_KERNELS_HPP='"d2d_kernels.hpp"' $flags -o differt2d_amd/csrc/libd2d.so differt2d_amd/csrc/d2d.hip 2>/dev/null || { echo "$tag build failed"; continue; }
  echo "== $tag"; bash scripts/sweep_bench.sh | grep -v "grid 256"
done
cp /tmp/libd2d_orig.so differt2d_amd/csrc/libd2d.so
