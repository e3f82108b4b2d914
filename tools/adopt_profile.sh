#!/bin/bash
# Copy the measurement set tools/profile_round.sh left in gpurun_out/<tag>/ into profiles/ (tracked) as <name>_* and make its PMC
# summary the one bench.py quotes (profiles/CURRENT_PMC):   tools/adopt_profile.sh <tag> <name, e.g. r05_b_wgrad>
set -eu
TAG=$1; NAME=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/gpurun_out/$TAG
for f in bench.json bench_under_rocprof.json kernel_stats.csv pmc_hbm_traffic.json pmc_hbm_traffic.txt pmc_mfma_busy.txt; do
    [ -s "$SRC/$f" ] && cp "$SRC/$f" "$ROOT/profiles/${NAME}_$f"
done
[ -s "$ROOT/profiles/${NAME}_pmc_hbm_traffic.json" ] && echo "${NAME}_pmc_hbm_traffic.json" > "$ROOT/profiles/CURRENT_PMC"
ls -la "$ROOT"/profiles/${NAME}_* "$ROOT/profiles/CURRENT_PMC"
