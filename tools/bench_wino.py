#!/usr/bin/env python3
"""Winograd 3x3 kernel: time vs reduction depth (Cin) at a fixed output — separates the per-chunk main-loop cost from the
fixed prologue / epilogue / launch cost.  Usage (GPU box): python tools/bench_wino.py [--hw 64 86] [--cout 256]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rpg_ramnet_amd import ops, _hip as H  # noqa: E402


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--hw", type=int, nargs=2, default=[64, 86])
    ap.add_argument("--cout", type=int, default=256)
    ap.add_argument("--epi", default="sigmoid")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B, (Hh, Ww), cout = a.batch, a.hw, a.cout
    taps = ops.Taps.get("conv", 3, 1)
    epi = {"sigmoid": H.EPI_SIGMOID, "relu": H.EPI_RELU, "linear": H.EPI_LINEAR}[a.epi]
    rows = []
    for cin in (32, 64, 128, 256, 512, 1024):
        w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=dev) * 0.05)
        b = torch.nn.Parameter(torch.randn(cout, device=dev) * 0.1)
        cp = ops.ConvParam([w], [b])
        x = torch.randn(B, Hh, Ww, cin, device=dev)
        y = torch.empty(B, Hh, Ww, cout, device=dev)
        t = {}
        for name, on, m24, split in (("wino", True, "off", False), ("direct", False, "off", False), ("w2x4", True, "force", False), ("split", True, "off", True)):
            ops.set_winograd(on)
            ops.set_winograd_2x4(m24)
            ops.set_winograd_split(split)
            t[name] = timeit(lambda: ops.conv_launch(x, taps, cp.fwd(), y, cout, bias=cp.bias(), epi=epi))
        ops.set_winograd(True)
        ops.set_winograd_2x4("auto")
        ops.set_winograd_split(True)
        gf = 2.0 * B * Hh * Ww * 9 * cin * cout / 1e9
        rows.append((cin, t, gf))
        print("Cin %4d  chunks %3d | F(2x2) %.3f ms (%6.1f TF/s eq.) | F(2x4) %.3f ms (%6.1f) | direct %.3f ms (%6.1f TF/s) | F(2x2) split reduction %.4f ms (unsplit %.4f)" %
              (cin, cin // 8, t["wino"], gf / t["wino"], t["w2x4"], gf / t["w2x4"], t["direct"], gf / t["direct"], t["split"], t["wino"]))
    (c0, t0, _), (c1, t1, _) = rows[1], rows[-1]
    for k in ("wino", "w2x4", "split"):
        per_chunk = (t1[k] - t0[k]) / ((c1 - c0) / 8)
        print("%-8s %.2f us per 8-channel chunk, fixed cost %.1f us" % (k, per_chunk * 1e3, (t0[k] - per_chunk * c0 / 8) * 1e3))


if __name__ == "__main__":
    main()
