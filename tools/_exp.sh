# Scratch script of the round's same-box A/B runs (boxes differ by ~3 %: all variants in ONE gpurun call):
#   gpurun -- 'bash tools/_exp.sh > gpurun_out/expNN.log 2>&1'
# This one: what the F(2x4) backward-weights kernel costs the co-scheduled step beyond its MFMAs (tools/abl_wgrad6.sh 1 7: timing builds, wrong gradients)
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])'
for i in 1 2; do
for L in librpg_ramnet_hip.so abl/lib_1.so abl/lib_7.so; do
echo "exact, $L"; RAMNET_HIP_LIB=rpg_ramnet_amd/$L python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "$P"
done
done
