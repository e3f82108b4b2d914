#!/usr/bin/env python3
"""Folded upsample-conv forward (ops._folded_upsample_conv): time of its parts vs the direct kernel.  GPU box: python tools/fold_parts.py"""
import sys, torch
sys.path.insert(0, "/root/repo")
from rpg_ramnet_amd import ops, _hip as H
from rpg_ramnet_amd.ops import _p, _st, Taps
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
dev = torch.device("cuda:0"); L = H.lib(); B = 8
for (cin, cout, Hh, W) in [(256, 128, 32, 43), (128, 64, 64, 86), (64, 32, 128, 172)]:
    w = torch.nn.Parameter(torch.randn(cout, cin, 5, 5, device=dev) * 0.05); b = torch.nn.Parameter(torch.randn(cout, device=dev))
    cp = ops.ConvParam([w], [b])
    x = torch.randn(B, Hh, W, cin, device=dev); s = torch.randn_like(x)
    y = torch.empty(B, 2 * Hh, 2 * W, cout, device=dev)
    xpad = torch.empty(B, Hh + 4, W + 4, cin, device=dev)
    t_pad = timeit(lambda: H.check(L.ramnet_pad2_sum(_p(x), _p(s), _p(xpad), B, Hh, W, cin, _st()), "pad"))
    fold = cp.pack_fold()
    classes = [(Taps.get("fold", 4, 0, py, px), Hh, W, (2, 2, py, px)) for py in range(2) for px in range(2)]
    t_all = timeit(lambda: ops._folded_upsample_conv(x, s, cp, y, H.EPI_RELU))
    t_dir = timeit(lambda: ops.conv_launch(x, Taps.get("conv", 5, 2), cp.fwd(), y, cout, x1=s, in_mode=H.IN_UP2X_SKIP, Hin=2 * Hh, Win=2 * W, bias=b, epi=H.EPI_RELU))
    a_rows = torch.empty(2, B * 2 * W, 5 * cin, device=dev); a_cols = torch.empty(2, B * 2 * Hh, 5 * cin, device=dev)
    t_im = timeit(lambda: H.check(L.ramnet_up2x_border_im2col(_p(x), _p(s), _p(a_rows), _p(a_cols), B, Hh, W, cin, _st()), "im2col"))
    wr, wc = cp.border_weights()
    t_mm = timeit(lambda: (torch.bmm(a_rows, wr), torch.bmm(a_cols, wc)))
    print("%3d->%3d %3dx%3d: pad %.3f  border im2col %.3f  border GEMMs %.3f  whole folded op %.3f  direct %.3f ms" % (cin, cout, Hh, W, t_pad, t_im, t_mm, t_all, t_dir))
