python -m pytest tests/test_hip_ops.py tests/test_hip_model.py tests/test_hip_graph.py tests/test_cabi_direct.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
echo HR0; RAMNET_GRU_HR=0 python bench.py --steps 10 --warmup 3 --resident-inputs --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['final_loss'])"
echo HR1; python bench.py --steps 10 --warmup 3 --resident-inputs --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['final_loss'])"
done
