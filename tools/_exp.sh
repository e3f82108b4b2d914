python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "wgrad or deferred or test_conv_gru or residual" 2>&1 | tail -3
run() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-timing --resident-inputs "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['final_loss'])"; }
echo xcd; run
echo noxcd; RAMNET_HIP_LIB=rpg_ramnet_amd/abl/lib_nox.so run
echo xcd; run
echo noxcd; RAMNET_HIP_LIB=rpg_ramnet_amd/abl/lib_nox.so run
echo xcd 384; run --wgrad-wino-blocks 384
echo xcd 256; run --wgrad-wino-blocks 256
python tools/bench_wgrad_defer.py --n 1 --reps 20 2>&1 | tail -6
RAMNET_HIP_LIB=rpg_ramnet_amd/abl/lib_nox.so python tools/bench_wgrad_defer.py --n 1 --reps 20 2>&1 | tail -6
