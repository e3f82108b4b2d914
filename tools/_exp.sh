python -m pytest tests/test_hip_properties.py -x -q -m gpu 2>&1 | grep -E "rel err|passed|failed" | head -5
python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "wgrad or deferred or test_conv_gru or residual or conv_layer" 2>&1 | tail -3
run() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-timing --resident-inputs "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['final_loss'])"; }
echo B new-layout; run
echo A old-layout+rfl; RAMNET_HIP_LIB=rpg_ramnet_amd/abl/lib_A.so run
echo C old-layout; RAMNET_HIP_LIB=rpg_ramnet_amd/abl/lib_C.so run
echo B new-layout; run
echo A old-layout+rfl; RAMNET_HIP_LIB=rpg_ramnet_amd/abl/lib_A.so run
echo B; python tools/bench_wgrad_defer.py --n 1 --reps 20 2>&1 | awk '/gru/{print $1, $2}' | tr '\n' ' '; echo
echo A; RAMNET_HIP_LIB=rpg_ramnet_amd/abl/lib_A.so python tools/bench_wgrad_defer.py --n 1 --reps 20 2>&1 | awk '/gru/{print $1, $2}' | tr '\n' ' '; echo
echo C; RAMNET_HIP_LIB=rpg_ramnet_amd/abl/lib_C.so python tools/bench_wgrad_defer.py --n 1 --reps 20 2>&1 | awk '/gru/{print $1, $2}' | tr '\n' ' '; echo
