#!/usr/bin/env python3
"""Stride-2 5x5 encoder layers: direct stride-2 kernels vs 3x3 Winograd over the space-to-depth input (ops.S2DConvParam),
forward / backward-data / backward-weights incl. the layout kernels.  GPU box: python tools/bench_s2d.py"""
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


dev = torch.device("cuda:0")
B = int(os.environ.get("S2D_B", "8"))
for cin, cout, Hh, W in [(32, 64, 256, 344), (64, 128, 128, 172), (128, 256, 64, 86)]:
    w = torch.nn.Parameter(torch.randn(cout, cin, 5, 5, device=dev) * 0.05)
    b = torch.nn.Parameter(torch.randn(cout, device=dev) * 0.1)
    cp = ops.ConvParam([w], [b])
    sp = cp.s2d()
    x = torch.randn(B, Hh, W, cin, device=dev)
    y = torch.empty(B, Hh // 2, W // 2, cout, device=dev)
    dy = torch.randn_like(y)
    dx = torch.empty_like(x)
    taps5, taps3, taps3d = ops.Taps.get("conv", 5, 2), ops.Taps.get("conv", 3, 1), ops.Taps.get("dgrad1", 3, 1)
    gf = 2.0 * B * (Hh // 2) * (W // 2) * 25 * cin * cout / 1e9
    # direct
    f0 = timeit(lambda: ops.conv_launch(x, taps5, cp.fwd(), y, cout, stride=2, bias=b, epi=H.EPI_RELU))
    d0 = timeit(lambda: ops.conv_launch_multi(y, cp.bwd(), dx, cin, [(ops.Taps.get("dgrad2", 5, 2, py, px), (Hh - py + 1) // 2, (W - px + 1) // 2, (2, 2, py, px))
                                                                   for py in range(2) for px in range(2)], xm=y, in_mode=H.IN_RELUMASK))
    ws = torch.zeros(25 * cin * cout, device=dev)
    bws = torch.zeros(cout, device=dev)
    g0 = timeit(lambda: ops.wgrad_launch(x, taps5, dy, ws, cout, stride=2, gmask=y, dbias=bws))
    # space-to-depth + Winograd
    xs = ops._space_to_depth(x)
    t_s2d = timeit(lambda: ops._space_to_depth(x))
    f1 = timeit(lambda: ops.conv_launch(xs, taps3, sp.fwd(), y, cout, bias=b, epi=H.EPI_RELU))
    gs = torch.empty_like(xs)
    d1 = timeit(lambda: ops.conv_launch(dy, taps3d, sp.bwd(), gs, 4 * cin, xm=y, in_mode=H.IN_RELUMASK))
    t_d2s = timeit(lambda: ops._space_to_depth(gs, inverse=True))
    ws3 = torch.zeros(16 * 4 * cin * cout, device=dev)
    ws3.wino = True
    g1 = timeit(lambda: ops.wgrad_launch(xs, taps3, dy, ws3, cout, gmask=y, dbias=bws))
    print("%3d->%3d %3dx%3d (%.1f GFLOP): fwd %.3f -> %.3f (+s2d %.3f)   dgrad %.3f -> %.3f (+d2s %.3f)   wgrad %.3f -> %.3f ms" %
          (cin, cout, Hh, W, gf, f0, f1, t_s2d, d0, d1, t_d2s, g0, g1))
    # the kernels addressing the space-to-depth view of x in place (RAMNET_IN_S2D loader / out_s2d epilogue): what the model runs
    tcs = ops.Taps.get("conv_s2d", 3, 1)
    f2 = timeit(lambda: ops.conv_launch(x, tcs, sp.fwd(), y, cout, in_mode=H.IN_S2D, Hin=Hh // 2, Win=W // 2, bias=b, epi=H.EPI_RELU))
    d2 = timeit(lambda: ops.conv_launch(dy, ops.Taps.get("dgrad1_s2d", 3, 1), sp.bwd(), dx, sp.Cin, xm=y, in_mode=H.IN_RELUMASK,
                                        Ho=Hh // 2, Wo=W // 2, out_s2d=cin))
    g2 = timeit(lambda: ops.wgrad_launch(x, tcs, dy, ws3, cout, in_mode=H.IN_S2D, Hin=Hh // 2, Win=W // 2, gmask=y, dbias=bws))
    print("        in place: fwd %.3f   dgrad %.3f   wgrad %.3f ms" % (f2, d2, g2))
