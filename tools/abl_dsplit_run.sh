for M in "$@"; do
  if [ $M = 0 ]; then L=rpg_ramnet_amd/librpg_ramnet_hip.so; else L=rpg_ramnet_amd/abl/libds_$M.so; fi
  echo "== mask $M"; RAMNET_HIP_LIB=$L timeout 120 python tools/bench_wgrad_dsplit.py 20 2>&1 | grep -v amdgpu.ids | sed -e 's/f2x4 [^|]*| //' 
done
