#!/bin/bash
# GPU box: instruction-cache / scalar-cache counters of the headline sweep kernel (two passes).   usage: pmc_icache.sh [approx]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
A=${1:-0}
CMD="python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras --approx $A"
rm -rf gpurun_out/pmc_ic1 gpurun_out/pmc_ic2 gpurun_out/pmc_ic3
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_BRANCH --output-format csv -d gpurun_out/pmc_ic1 -- $CMD > gpurun_out/pmc_ic1.log 2>&1
rocprofv3 --kernel-trace --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ SQC_TC_STALL --output-format csv -d gpurun_out/pmc_ic2 -- $CMD > gpurun_out/pmc_ic2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH_LEVEL SQ_INST_LEVEL_SMEM SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_ic3 -- $CMD > gpurun_out/pmc_ic3.log 2>&1
for d in 1 2 3; do python3 scripts/pmc_kernel.py gpurun_out/pmc_ic$d "power_fwd_kernel<${A}, false, 2, false, true, 1>"; done
