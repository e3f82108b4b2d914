#!/usr/bin/env python3
"""Which torch (ATen) kernels a training step of the bench shape still launches, and from where: torch.profiler over ONE step (resident
inputs), device kernels that are not the library's grouped by name and by the Python call site that issued them.
GPU box: python tools/aten_glue.py [--top 30]"""
import argparse
import collections
import contextlib
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=30)
    a = ap.parse_args()
    from rpg_ramnet_amd import ops
    from rpg_ramnet_amd.model.model import ERGB2DepthRecurrent
    from rpg_ramnet_amd.parallel import FlatGradReducer
    from rpg_ramnet_amd.trainer import sequence_loss
    ops.set_wgrad_overlap(True)
    ops.set_decoder_overlap(True)
    K, bins, B, L, H, W = 5, 5, 8, 8, 256, 344
    cfg = dict(bench.RELEASED, num_bins_events=bins, gpu=0, every_x_rgb_frame=K, baseline=False, loss_composition=["image", "events4"],
               state_combination="convgru")
    torch.manual_seed(0)
    with contextlib.redirect_stdout(sys.stderr):
        model = ERGB2DepthRecurrent(cfg)
    model = model.to(model.gpu).train()
    seq = bench.synth_sequence(model, B, L, H, W, K, bins, 200000, seed=1000)
    reducer = FlatGradReducer(model)
    opt = torch.optim.Adam(model.parameters(), lr=3e-4, weight_decay=0, fused=True)

    def step():
        reducer.zero()
        total, _ = sequence_loss(model, seq, cfg["loss_composition"], [1, 1])
        total.backward()
        reducer.all_reduce()
        reducer.wait()
        opt.step()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True, experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
        step()
        torch.cuda.synchronize()
    by_op = collections.Counter()
    by_site = collections.Counter()
    dur = collections.Counter()
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for ev in prof.events():
        if ev.device_type != torch.autograd.DeviceType.CPU or not ev.kernels:
            continue
        if not ev.name.startswith("aten::"):
            continue
        site = "?"
        frames = [fr for fr in (ev.stack or []) if ("rpg_ramnet_amd" in fr or "bench.py" in fr or "aten_glue" in fr) and "/torch/" not in fr]
        if frames:
            site = " <- ".join(fr.replace(here + "/", "").replace("rpg_ramnet_amd/", "") for fr in frames[:3])
        shape = ""
        try:
            shape = str([list(x) for x in (ev.input_shapes or []) if x][:2])
        except Exception:      # noqa: BLE001
            pass
        site = site + " " + shape
        n = len(ev.kernels)
        d = sum(k.duration for k in ev.kernels)
        by_op[ev.name] += n
        by_site[(ev.name, site)] += n
        dur[(ev.name, site)] += d
    print("ATen ops that launched device kernels in one step: %d launches" % sum(by_op.values()))
    for k, v in by_op.most_common(12):
        print("  %-28s %5d" % (k, v))
    print("by call site (launches, total device us):")
    for (op, site), v in by_site.most_common(a.top):
        print("  %5d %9.1f  %-24s %s" % (v, dur[(op, site)], op, site))


if __name__ == "__main__":
    main()
