q() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d.get('value'), d.get('ms_per_step'), str(d.get('input_side'))[:40], d['config']['workload'][:60])"; }
echo full-frame; python bench.py --steps 3 --warmup 1 --no-cpu-baseline --full-frame 2>/dev/null | q
echo convlstm; python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --state convlstm 2>/dev/null | q
echo infer; python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --mode infer 2>/dev/null | q
echo stream; python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --mode stream --batch 1 2>/dev/null | q
echo gpus2; python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --batch 2 --seq-len 2 --events-per-grid 20000 2>/dev/null | q
echo graph; python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --graph 2>/dev/null | q
echo configs4; python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --height 480 --width 640 --bins 10 --batch 4 --seq-len 16 2>/dev/null | q
