for A in "" "--wgrad-wino-blocks 320" "--wgrad-wino-blocks 256" "--wgrad-defer 5" "--time-batch-max-decodes 3" "--no-gru-bwd-fused" "" ; do
  echo "== $A"; python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-timing $A 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['final_loss'])"
done
