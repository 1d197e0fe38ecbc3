#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + PMC passes of the bench command.
# Writes under gpurun_out/prof_*; copy the summaries you want judged into profiles/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out
TAG=${1:-r02}
APPROX=${2:-0}
CMD="python3 bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-extras --approx ${APPROX}"
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_a${APPROX}_trace -- $CMD > $OUT/prof_${TAG}_a${APPROX}_trace.log 2>&1
# PMC passes, each on its own (never combined with trace domains other than kernel-trace)
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/prof_${TAG}_a${APPROX}_pmc1 -- $CMD > $OUT/prof_${TAG}_a${APPROX}_pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/prof_${TAG}_a${APPROX}_pmc2 -- $CMD > $OUT/prof_${TAG}_a${APPROX}_pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_a${APPROX}_pmc3 -- $CMD > $OUT/prof_${TAG}_a${APPROX}_pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}_a${APPROX}_pmc4 -- $CMD > $OUT/prof_${TAG}_a${APPROX}_pmc4.log 2>&1
find $OUT -name "*.csv" | head -40
