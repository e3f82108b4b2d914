#!/bin/bash
# Full measurement set of one build on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag> [bench steps] [bench warmup]
# -> gpurun_out/<tag>/: bench.json (un-profiled), kernel_stats.csv + bench_under_rocprof.json (rocprofv3 --kernel-trace --stats of
#    the same command), pmc_hbm_traffic.{json,txt} (separate --pmc FETCH_SIZE / WRITE_SIZE passes), pmc_mfma_busy.txt.
# Copy what should be judged into profiles/ (tracked) as rNN_<tag>_*.
set -u
TAG=${1:-run}; STEPS=${2:-5}; WARM=${3:-2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/bench.py --steps $STEPS --warmup $WARM > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; tail -c 400 $OUT/bench.json
rm -rf /tmp/prof_stats && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o s -- \
    python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
cp $(find /tmp/prof_stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv 2>/dev/null
for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$C && timeout 1200 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$C -o c -- \
        python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing --no-extras > $OUT/pmc_$C.log 2>&1
done
python $ROOT/tools/pmc_traffic.py $(find /tmp/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1) \
    $(find /tmp/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1) $OUT/pmc_hbm_traffic.json > $OUT/pmc_hbm_traffic.txt 2>&1
rm -rf /tmp/pmc_mfma && timeout 1200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv \
    -d /tmp/pmc_mfma -o c -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing --no-extras > $OUT/pmc_mfma.log 2>&1
python $ROOT/tools/pmc_summary.py $(find /tmp/pmc_mfma -name '*counter_collection.csv' | head -1) > $OUT/pmc_mfma_busy.txt 2>&1
ls -la $OUT
