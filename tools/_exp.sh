for i in 1 2; do
echo OLD; RAMNET_HIP_LIB=rpg_ramnet_amd/abl/lib_old.so python bench.py --steps 10 --warmup 3 --resident-inputs --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
echo NEW; python bench.py --steps 10 --warmup 3 --resident-inputs --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
RAMNET_HIP_LIB=rpg_ramnet_amd/abl/lib_probe6.so python tools/probe_wino6.py 2>&1 | grep -v "^$" | grep -v "alive\|first 512\|later ones"
