python -m pytest tests/test_hip_ops.py -x -q -k "test_winograd2x4_conv3x3_raw or (winograd2x4 and conv_gru)" 2>&1 | tail -2
for R in 1 2; do
for LIB in "" "rpg_ramnet_amd/abl/lib_r6_old.so"; do
  echo "== lib=$LIB"; RAMNET_HIP_LIB=$LIB python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['final_loss'])"
done
done
