run() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-timing --resident-inputs "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['final_loss'])"; }
echo prologue2; run
echo prologue1; RAMNET_HIP_LIB=rpg_ramnet_amd/abl/lib_p1.so run
echo prologue2; run
echo prologue1; RAMNET_HIP_LIB=rpg_ramnet_amd/abl/lib_p1.so run
python tools/bench_layers.py --only gru 2>&1 | tail -8
RAMNET_HIP_LIB=rpg_ramnet_amd/abl/lib_p1.so python tools/bench_layers.py --only gru 2>&1 | tail -8
