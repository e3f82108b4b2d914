python -m pytest tests/test_hip_ops.py tests/test_hip_properties.py -x -q -m gpu -k "wgrad or weights or backward or grad" 2>&1 | tail -3
echo OLD; RAMNET_HIP_LIB=rpg_ramnet_amd/abl/lib_old.so python tools/bench_wgrad_defer.py --n 1 2>/dev/null
echo NEW; python tools/bench_wgrad_defer.py --n 1 2>/dev/null
for i in 1 2; do
echo OLD; RAMNET_HIP_LIB=rpg_ramnet_amd/abl/lib_old.so python bench.py --steps 10 --warmup 3 --resident-inputs --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
echo NEW; python bench.py --steps 10 --warmup 3 --resident-inputs --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
