# Scratch script of the round's same-box A/B runs (boxes differ by ~3 %: both variants in ONE gpurun call):
#   gpurun -- 'bash tools/_exp.sh > gpurun_out/expNN.log 2>&1'
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["final_loss"])'
for i in 1 2; do
echo "split, F(2x4) backward-weights"; RAMNET_SPLIT_WGRAD=0 python bench.py --steps 10 --warmup 3 --split-operands --no-extras --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "$P"
echo "split, direct split backward-weights"; RAMNET_SPLIT_WGRAD=1 python bench.py --steps 10 --warmup 3 --split-operands --no-extras --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "$P"
done
