#!/bin/bash
for rep in 1 2; do
for v in 1 0; do
  RAMNET_AB=$v timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ab=$v train', round(d['value'],2), round(d['ms_per_step'],2), d['final_loss'])"
done
done
