set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/m
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$C && timeout 1200 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$C -o c -- \
        python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing --no-extras > $OUT/pmc_$C.log 2>&1
done
python $ROOT/tools/pmc_traffic.py $(find /tmp/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1) \
    $(find /tmp/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1) $OUT/pmc_hbm_traffic.json > $OUT/pmc_hbm_traffic.txt 2>&1
grep "wino_r_kernel<2>" $OUT/pmc_hbm_traffic.txt
