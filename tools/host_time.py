#!/usr/bin/env python3
"""Host-side issue time of a training step vs its GPU time: the step is timed phase by phase WITHOUT synchronising (how long the
host needs to queue the forward, the backward and the optimizer while the GPU still works on earlier launches) and once with
fences.  If the un-fenced issue time approaches the fenced step time, the launch path (Python + ctypes + autograd) is the
bottleneck, not the kernels.  GPU box: python tools/host_time.py [--steps 4]"""
import argparse
import contextlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--fused-adam", type=int, default=0)
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
    opt = torch.optim.Adam(model.parameters(), lr=3e-4, weight_decay=0, fused=bool(a.fused_adam))

    def step(stamps=None):
        t0 = time.perf_counter()
        reducer.zero()
        total, _ = sequence_loss(model, seq, cfg["loss_composition"], [1, 1])
        t1 = time.perf_counter()
        total.backward()
        t2 = time.perf_counter()
        reducer.all_reduce()
        reducer.wait()
        opt.step()
        t3 = time.perf_counter()
        if stamps is not None:
            stamps.append((t1 - t0, t2 - t1, t3 - t2))

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    # fenced: GPU time per step
    t = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    gpu = (time.perf_counter() - t) / a.steps
    # un-fenced issue times, GPU kept busy by the previous steps
    st = []
    for _ in range(a.steps):
        step(st)
    torch.cuda.synchronize()
    f, b, o = [sum(x[i] for x in st) / len(st) for i in range(3)]
    print("step (fenced over %d steps): %.1f ms" % (a.steps, 1e3 * gpu))
    print("host issue time per step, not synchronised: forward+loss %.1f ms, backward %.1f ms, all-reduce+Adam %.2f ms, sum %.1f ms"
          % (1e3 * f, 1e3 * b, 1e3 * o, 1e3 * (f + b + o)))
    print("(the issue times include waiting on a full launch queue: they are an upper bound of the host work)")
    # host work alone: the same step with the GPU idle at the start of every phase
    st2 = []
    for _ in range(2):
        torch.cuda.synchronize()
        step(st2)
    torch.cuda.synchronize()
    f, b, o = [sum(x[i] for x in st2) / len(st2) for i in range(3)]
    print("host issue time per step from an idle GPU: forward+loss %.1f ms, backward %.1f ms, all-reduce+Adam %.2f ms" % (1e3 * f, 1e3 * b, 1e3 * o))


if __name__ == "__main__":
    main()
