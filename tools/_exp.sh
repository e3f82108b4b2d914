run() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-timing --resident-inputs "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['final_loss'])"; }
echo base; run
echo side2; RAMNET_SIDE2=1 run
echo base; run
echo side2; RAMNET_SIDE2=1 run
python tools/host_profile.py 2>/dev/null | head -3
