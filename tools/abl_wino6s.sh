#!/bin/bash
# Ablation builds of csrc/conv_wino6s.hip (RAMNET_ABL6S masks, see the file) -> rpg_ramnet_amd/abl/lib6s_<mask>.so (git-ignored, travels with
# gpurun); time with:  tools/abl_wino6s_run.sh <masks>   (= RAMNET_HIP_LIB=rpg_ramnet_amd/abl/lib6s_<mask>.so python tools/bench_split_operands.py --quick)
# (mask 128 = the full loop in the same reduced build: the reference point)
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$ROOT/rpg_ramnet_amd
mkdir -p $PKG/abl
python -c "from rpg_ramnet_amd import build; build.build()" > /dev/null
OBJS=$(ls $PKG/build/*.o | grep -v conv_wino6s.o)
for M in "$@"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DRAMNET_ABL6S=$M -c $PKG/csrc/conv_wino6s.hip -o /tmp/w6s_abl_$M.o &
done
wait
for M in "$@"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/abl/lib6s_$M.so $OBJS /tmp/w6s_abl_$M.o
done
ls -la $PKG/abl
