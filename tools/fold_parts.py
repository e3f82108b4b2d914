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
    side_buf = torch.zeros(B, 2 * Hh, 4, cout, device=dev)
    def main():
        ops.conv_launch_multi(xpad, fold, y, cout, classes, bias=cp.bias(), epi=H.EPI_RELU, beta=1.0, frame=2, e0=side_buf)
    t_all = timeit(lambda: ops._folded_upsample_conv(x, s, cp, y, H.EPI_RELU))
    t_dir = timeit(lambda: ops.conv_launch(x, Taps.get("conv", 5, 2), cp.fwd(), y, cout, x1=s, in_mode=H.IN_UP2X_SKIP, Hin=2 * Hh, Win=2 * W, bias=b, epi=H.EPI_RELU))
    print("%3d->%3d %3dx%3d: pad %.3f  main %.3f  whole folded op %.3f  direct %.3f ms" % (cin, cout, Hh, W, t_pad, timeit(main), t_all, t_dir))
    neg = cp.pack_neg()
    H2, W2 = 2 * Hh, 2 * W
    plan = [((3, W2 + 4), "band_rows_lo", y, (2, W2), (0, 0)), ((3, W2 + 4), "band_rows_hi", y, (2, W2), (H2 - 2, 0)),
            ((H2, 3), "band_cols_lo", side_buf, (H2, 2), (0, 0)), ((H2, 3), "band_cols_hi", side_buf, (H2, 2), (0, 2))]
    bands = [torch.randn(B, bh, bw, cin, device=dev) for (bh, bw), _, _, _, _ in plan]
    ts = []
    for side, (_, kind, target, (Ho, Wo), (oy, ox)) in enumerate(plan):
        ts.append(timeit(lambda: ops.conv_launch(bands[side], Taps.get(kind, 5, 2), neg, target, cout, Ho=Ho, Wo=Wo, os=(1, 1, oy, ox), epi=H.EPI_LINEAR)))
    tk = timeit(lambda: [H.check(L.ramnet_up2x_ring_band(_p(x), _p(s), _p(bands[i]), B, Hh, W, cin, i, _st()), "band") for i in range(4)])
    print("          band launches alone (ms): rows %.3f %.3f  cols %.3f %.3f   band kernels x4 %.3f" % (ts[0], ts[1], ts[2], ts[3], tk))
    import time
    def conc():
        ready = torch.cuda.current_stream().record_event()
        for side, (_, kind, target, (Ho, Wo), (oy, ox)) in enumerate(plan):
            st = ops._band_stream(dev, side)
            st.wait_event(ready)
            with torch.cuda.stream(st):
                ops.conv_launch(bands[side], Taps.get(kind, 5, 2), neg, target, cout, Ho=Ho, Wo=Wo, os=(1, 1, oy, ox), epi=H.EPI_LINEAR)
            torch.cuda.current_stream().wait_event(st.record_event())
    conc(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): conc()
    torch.cuda.synchronize()
    print("          4 band launches on 4 streams: %.3f ms per round (wall)" % ((time.perf_counter() - t0) / 20 * 1e3))
