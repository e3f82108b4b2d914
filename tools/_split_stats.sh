cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/ps && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o s -- python $R/bench.py --split-operands --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/r06_g_split_bench_under_rocprof.json 2> $R/gpurun_out/r06_g_rocprof.err
cp $(find /tmp/ps -name '*kernel_stats.csv' | head -1) $R/gpurun_out/r06_g_split_kernel_stats.csv
head -25 $R/gpurun_out/r06_g_split_kernel_stats.csv
