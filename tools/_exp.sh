# Scratch script of the round's same-box A/B runs (boxes differ by ~3 %: both builds in ONE gpurun call):
#   cp rpg_ramnet_amd/librpg_ramnet_hip.so rpg_ramnet_amd/abl/lib_old.so   # before rebuilding
#   gpurun -- 'bash tools/_exp.sh > gpurun_out/expNN.log 2>&1'
for i in 1 2; do
echo OLD; RAMNET_HIP_LIB=rpg_ramnet_amd/abl/lib_old.so python bench.py --steps 10 --warmup 3 --resident-inputs --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
echo NEW; python bench.py --steps 10 --warmup 3 --resident-inputs --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
