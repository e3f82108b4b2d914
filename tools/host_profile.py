#!/usr/bin/env python3
"""cProfile of the HOST side of one training step (bench shape): where the Python launch path spends its time.
GPU box: python tools/host_profile.py [--top 40]"""
import argparse
import contextlib
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=45)
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
    t = time.perf_counter()
    step()
    host = time.perf_counter() - t
    torch.cuda.synchronize()
    print("host issue time of one step from an idle GPU: %.1f ms" % (1e3 * host))
    pr = cProfile.Profile()
    pr.enable()
    step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr, stream=sys.stdout)
    st.sort_stats("tottime").print_stats(a.top)


if __name__ == "__main__":
    main()
