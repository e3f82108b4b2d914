#!/usr/bin/env python3
"""Does training with split bf16 operands follow the exact-fp32 trajectory?  The bench's training step (B = 8, L = 8, K = 5, 256 x 344, Adam lr
3e-4, seeded weights, two alternating synthetic sequences) for N optimizer steps with the exact F(2x4,3x3) kernels and again with
ops.set_split_operands(True): per-step losses side by side, their largest relative difference, and the largest weight difference at the end
against the distance the weights travelled — and, for scale, the same two numbers between two exact-fp32 algorithms (F(2x4) vs F(2x2)).   Usage (GPU box): python tools/split_trajectory.py [--steps 40]"""
import argparse
import contextlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from rpg_ramnet_amd import ops  # noqa: E402
from rpg_ramnet_amd.model.model import ERGB2DepthRecurrent  # noqa: E402
from rpg_ramnet_amd.trainer import sequence_loss  # noqa: E402


def run(split, steps, seqs, cfg, w2x4="auto"):
    ops.set_split_operands(split)
    ops.set_winograd_2x4(w2x4)
    torch.manual_seed(0)
    with contextlib.redirect_stdout(sys.stderr):
        model = ERGB2DepthRecurrent(cfg)
    model = model.to(model.gpu).train()
    w0 = {k: p.detach().clone() for k, p in model.named_parameters()}
    opt = torch.optim.Adam(model.parameters(), lr=3e-4, weight_decay=0, fused=True)
    losses = []
    for it in range(steps):
        opt.zero_grad(set_to_none=False)
        total, _ = sequence_loss(model, seqs[it & 1], cfg["loss_composition"], [1, 1])
        total.backward()
        opt.step()
        losses.append(float(total.detach()))
    ops.set_split_operands(False)
    ops.set_winograd_2x4("auto")
    return losses, {k: p.detach().clone() for k, p in model.named_parameters()}, w0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    a = ap.parse_args()
    ops.set_wgrad_overlap(True)
    ops.set_decoder_overlap(True)
    cfg = dict(bench.RELEASED, num_bins_events=5, gpu=0, every_x_rgb_frame=5, baseline=False, loss_composition=["image", "events4"], state_combination="convgru")
    torch.manual_seed(0)
    with contextlib.redirect_stdout(sys.stderr):
        m0 = ERGB2DepthRecurrent(cfg)
    m0 = m0.to(m0.gpu)
    seqs = [bench.synth_sequence(m0, 8, 8, 256, 344, 5, 5, 200000, seed=1000 + i) for i in range(2)]
    del m0
    le, we, w0 = run(False, a.steps, seqs, cfg)
    ls, ws, _ = run(True, a.steps, seqs, cfg)
    l2, w2, _ = run(False, a.steps, seqs, cfg, w2x4="off")          # a second EXACT-fp32 algorithm (F(2x2,3x3) everywhere) for scale
    print("step   exact fp32      split operands   rel. diff")
    worst = 0.0
    for i, (x, y) in enumerate(zip(le, ls)):
        d = abs(x - y) / abs(x)
        worst = max(worst, d)
        if i < 5 or i % 5 == 4:
            print("%4d   %.7f      %.7f       %.1e" % (i, x, y, d))
    dw = max(float((we[k] - ws[k]).abs().max()) for k in we)
    travel = max(float((we[k] - w0[k]).abs().max()) for k in we)
    print("largest relative loss difference over %d steps: %.2e; loss %.5f -> %.5f" % (a.steps, worst, le[0], le[-1]))
    print("largest weight difference exact vs split after %d steps: %.2e (the weights moved by up to %.2e)" % (a.steps, dw, travel))
    d2 = max(float((we[k] - w2[k]).abs().max()) for k in we)
    print("for scale — two EXACT fp32 algorithms, F(2x4,3x3) vs F(2x2,3x3): largest relative loss difference %.2e, largest weight difference %.2e "
          "(Adam normalises a gradient that is rounding noise around zero to a full-size step of either sign)" % (
              max(abs(x - y) / abs(x) for x, y in zip(le, l2)), d2))


if __name__ == "__main__":
    main()
