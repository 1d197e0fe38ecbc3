# usage: ab_run.sh "<extra hipcc flags>" <command...>   (GPU box: runs the command against a variant build, then restores)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp differt2d_amd/csrc/libd2d.so /tmp/libd2d_orig.so
flags="$1"; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -DD2D_KERNELS_HPP='"d2d_kernels.hpp"' $flags -o differt2d_amd/csrc/libd2d.so differt2d_amd/csrc/d2d.hip 2>/dev/null
"$@"
cp /tmp/libd2d_orig.so differt2d_amd/csrc/libd2d.so
