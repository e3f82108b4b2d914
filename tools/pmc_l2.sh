#!/bin/bash
# L2 hit rate and memory-side read requests per kernel of one training step (which operand is re-fetched):
#   tools/pmc_l2.sh <tag>  -> gpurun_out/<tag>/pmc_l2.txt      (run through gpurun from the repo root)
set -u
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_l2 && timeout 1200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d /tmp/pmc_l2 -o c -- \
    python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing --no-extras --resident-inputs > $OUT/pmc_l2.log 2>&1
python - "$(find /tmp/pmc_l2 -name '*counter_collection.csv' | head -1)" > $OUT/pmc_l2.txt <<'P'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"].split("(")[0].replace("void ramnet::", "").replace("ramnet::", ""), r["Grid_Size"])
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "TCC_HIT_sum":
        n[k] += 1
print("%-58s %9s %6s | %12s %12s %7s | %12s %10s" % ("kernel", "grid", "calls", "TCC_HIT", "TCC_MISS", "hit %", "EA_RDREQ", "MB/launch"))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("TCC_MISS_sum", 0)):
    h, m, rd, rd32 = v.get("TCC_HIT_sum", 0), v.get("TCC_MISS_sum", 0), v.get("TCC_EA0_RDREQ_sum", 0), v.get("TCC_EA0_RDREQ_32B_sum", 0)
    if n[k] == 0 or h + m == 0:
        continue
    mb = ((rd - rd32) * 64 + rd32 * 32) / n[k] / 1e6
    print("%-58s %9s %6d | %12.4g %12.4g %6.1f%% | %12.4g %10.1f" % (k[0][:58], k[1], n[k], h / n[k], m / n[k], 100 * h / (h + m), rd / n[k], mb))
P
head -30 $OUT/pmc_l2.txt
