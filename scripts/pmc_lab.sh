#!/bin/bash
# GPU box: PMC counters of the sweep kernel of one opt_lab setting.   usage: pmc_lab.sh <tag> <kernel substring> <opt_lab args...>
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
tag=$1; kern=$2; shift 2
rm -rf gpurun_out/pmc_$tag
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_$tag -- python3 scripts/opt_lab.py "$@" > gpurun_out/pmc_$tag.log 2>&1
echo "== $tag: $*"
python3 scripts/pmc_kernel.py gpurun_out/pmc_$tag "$kern"
