#!/bin/bash
# configs[3] (batch-1 streaming) measurement set on the GPU box:  tools/profile_stream.sh <tag>
# -> gpurun_out/<tag>/: stream_bench.json (un-profiled), stream_kernel_stats.csv + stream_kernel_trace.csv (rocprofv3 --kernel-trace
#    --stats of the same command; the graph replays dispatch the same kernels), stream_chain.txt (per-kernel table of one update).
set -u
TAG=${1:-stream}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/bench.py --mode stream --batch 1 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/stream_bench.json 2> $OUT/stream_bench.err
echo "stream bench rc=$?"; head -c 600 $OUT/stream_bench.json; echo
rm -rf /tmp/prof_stream && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stream -o s -- \
    python $ROOT/bench.py --mode stream --batch 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timing > $OUT/stream_bench_under_rocprof.json 2> $OUT/stream_rocprof.err
cp $(find /tmp/prof_stream -name '*kernel_stats.csv' | head -1) $OUT/stream_kernel_stats.csv 2>/dev/null
python $ROOT/tools/stream_chain.py $(find /tmp/prof_stream -name '*kernel_trace.csv' | head -1) > $OUT/stream_chain.txt 2>&1
head -50 $OUT/stream_chain.txt
