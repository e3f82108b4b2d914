python -m pytest tests/test_hip_ops.py tests/test_hip_fullsize.py -x -q -m gpu -k "wgrad or deferred or test_conv_gru or residual or conv_layer or encoder or full_size or B8_L8" 2>&1 | tail -3
run() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-timing --resident-inputs "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['final_loss'])"; }
echo new-sched; run
echo old-sched; RAMNET_HIP_LIB=rpg_ramnet_amd/abl/lib_S.so run
echo new-sched; run
echo old-sched; RAMNET_HIP_LIB=rpg_ramnet_amd/abl/lib_S.so run
echo new; python tools/bench_wgrad_defer.py --n 1 --reps 20 2>&1 | awk '/gru/{print $1, $2}' | tr '\n' ' '; echo
echo old; RAMNET_HIP_LIB=rpg_ramnet_amd/abl/lib_S.so python tools/bench_wgrad_defer.py --n 1 --reps 20 2>&1 | awk '/gru/{print $1, $2}' | tr '\n' ' '; echo
