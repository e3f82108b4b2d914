#!/bin/bash
# Time the ablation builds of csrc/conv_wino6s.hip that tools/abl_wino6s.sh left in rpg_ramnet_amd/abl/ (run through gpurun from the repo root):
#   tools/abl_wino6s_run.sh 128 1 2 3 95       (mask 128 = the full loop in the same reduced build: the reference point)
for M in "$@"; do echo "== mask $M"; RAMNET_HIP_LIB=rpg_ramnet_amd/abl/lib6s_$M.so python tools/bench_split_operands.py --quick 2>&1 | grep -v amdgpu.ids; done
