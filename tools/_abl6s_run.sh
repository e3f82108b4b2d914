for M in 128 384 95 351; do echo "== mask $M"; RAMNET_HIP_LIB=rpg_ramnet_amd/abl/lib6s_$M.so python tools/bench_split_operands.py --quick 2>&1 | grep -v amdgpu.ids; done
