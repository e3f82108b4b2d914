#!/bin/bash
# Ablation builds of csrc/conv_wgrad_wino6.hip (RAMNET_ABL masks, see the file) -> rpg_ramnet_amd/abl/lib_<mask>.so (git-ignored, travels with
# gpurun); time with:  RAMNET_HIP_LIB=rpg_ramnet_amd/abl/lib_<mask>.so python tools/bench_wgrad_defer.py --n 1
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$ROOT/rpg_ramnet_amd
mkdir -p $PKG/abl
python -c "from rpg_ramnet_amd import build; build.build()" > /dev/null
OBJS=$(ls $PKG/build/*.o | grep -v conv_wgrad_wino6.o)
for M in "$@"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DRAMNET_ABL=$M -c $PKG/csrc/conv_wgrad_wino6.hip -o /tmp/w6_abl_$M.o &
done
wait
for M in "$@"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/abl/lib_$M.so $OBJS /tmp/w6_abl_$M.o
done
ls -la $PKG/abl
