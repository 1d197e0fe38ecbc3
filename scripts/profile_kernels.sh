#!/bin/bash
# GPU box (via gpurun): rocprofv3 kernel stats + PMC passes (the counter sets of scripts/profile_gpu.sh, each on its own) of
# every case of scripts/kernel_lab.py, under gpurun_out/<tag>_kernels/ (scripts/summarize_kernels.py condenses them into profiles/).
#   usage: profile_kernels.sh [tag] [cases...]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${1:-r03}; shift || true
CASES=${*:-"cfg3 cfg3_hsig txg cfg4 sigmoid cfg5 cfg5_fwd"}
OUT=gpurun_out/${TAG}_kernels
mkdir -p $OUT
run() {  # name, rocprofv3 arguments ...
  local name=$1; shift
  rm -rf $OUT/$name
  timeout -k 10 150 rocprofv3 "$@" --output-format csv -d $OUT/$name -- python3 scripts/kernel_lab.py $c $n > $OUT/$name.log 2>&1
  echo "$name rc=$?"
}
for c in $CASES; do
  n=20; [ "$c" = "sigmoid" ] && n=6; [ "$c" = "cfg5_tan" ] && n=4
  run ${c}_trace --kernel-trace --stats
  run ${c}_pmc1 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
  run ${c}_pmc2 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
  run ${c}_pmc3 --kernel-trace --pmc FETCH_SIZE
  run ${c}_pmc4 --kernel-trace --pmc WRITE_SIZE
  grep -h "sweep kernel" $OUT/${c}_trace.log
done
