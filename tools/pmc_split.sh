#!/bin/bash
# PMC pass over tools/bench_split_operands.py: MFMA-busy, wave cycles, wait buckets and LDS conflicts of the exact-fp32 F(2x4,3x3) launches of a ConvGRU
# update (conv_wino_r6_kernel) and of their split-operand forms (conv_wino_r6s_kernel), per launch shape.   tools/pmc_split.sh <tag>
#   -> gpurun_out/<tag>/pmc_split.txt       (run through gpurun from the repo root; counters in a pass of their own: --kernel-trace + --pmc only)
set -u
TAG=${1:-s}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_s && timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS \
    --output-format csv -d /tmp/pmc_s -o c -- python $ROOT/tools/bench_split_operands.py --reps 5 > $OUT/pmc_split.log 2>&1
python $ROOT/tools/pmc_wino.py $(find /tmp/pmc_s -name '*counter_collection.csv' | head -1) $(find /tmp/pmc_s -name '*kernel_trace.csv' | head -1) > $OUT/pmc_split.txt 2>&1
cat $OUT/pmc_split.txt
