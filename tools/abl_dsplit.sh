#!/bin/bash
# Ablation builds of csrc/conv_wgrad_dsplit.hip (RAMNET_ABLD masks, see the file) -> rpg_ramnet_amd/abl/libds_<mask>.so (git-ignored, travels with
# gpurun); time with:  RAMNET_HIP_LIB=rpg_ramnet_amd/abl/libds_<mask>.so python tools/bench_wgrad_dsplit.py
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$ROOT/rpg_ramnet_amd
mkdir -p $PKG/abl
python -c "from rpg_ramnet_amd import build; build.build()" > /dev/null
OBJS=$(ls $PKG/build/*.o | grep -v conv_wgrad_dsplit.o)
for M in "$@"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DRAMNET_ABLD=$M ${DSFLAGS:-} -c $PKG/csrc/conv_wgrad_dsplit.hip -o /tmp/ds_abl_$M.o &
done
wait
for M in "$@"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/abl/libds_$M.so $OBJS /tmp/ds_abl_$M.o
done
ls $PKG/abl
