#!/bin/bash
# PMC pass over tools/bench_wino.py (one map size): MFMA-busy, wave cycles and wait buckets per kernel instantiation.
#   tools/pmc_wino.sh <tag> <H> <W> <cout>     -> gpurun_out/<tag>/pmc_wino.txt
set -u
TAG=${1:-w}; H=${2:-128}; W=${3:-172}; CO=${4:-128}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_w && timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS \
    --output-format csv -d /tmp/pmc_w -o c -- python $ROOT/tools/bench_wino.py --hw $H $W --cout $CO > $OUT/pmc_wino.log 2>&1
python $ROOT/tools/pmc_wino.py $(find /tmp/pmc_w -name '*counter_collection.csv' | head -1) $(find /tmp/pmc_w -name '*kernel_trace.csv' | head -1) > $OUT/pmc_wino_${H}x${W}x${CO}.txt 2>&1
cat $OUT/pmc_wino_${H}x${W}x${CO}.txt
