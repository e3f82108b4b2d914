#!/usr/bin/env python3
"""Backward-weights of n ConvGRU cell updates: n launches against ONE multi-segment launch (ramnet_wgrad_desc.segs), per scale, in stream
order on an otherwise idle chip.  Usage (GPU box): python tools/bench_wgrad_defer.py [--n 5] [--batch 8]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rpg_ramnet_amd import ops, _hip as H  # noqa: E402


def timeit(fn, reps):
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
    ap.add_argument("--n", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--blocks", type=int, default=0)
    a = ap.parse_args()
    if a.blocks:
        H.check(H.lib().ramnet_set_option(b"wgrad_wino_blocks", a.blocks), "set_option")
    dev = torch.device("cuda:0")
    ops.set_wgrad_winograd_2x4("force")
    taps = ops.Taps.get("conv", 3, 1)
    print("%-10s %10s %10s %10s" % ("layer", "n launches", "1 launch", "ratio"))
    for i, C in enumerate([64, 128, 256]):
        Hh, Ww = 256 >> (i + 1), 344 >> (i + 1)
        for kind, cout in (("ur", 2 * C), ("o", C)):
            w = torch.nn.Parameter(torch.randn(cout, 2 * C, 3, 3, device=dev) * 0.01)
            b = torch.nn.Parameter(torch.zeros(cout, device=dev))
            cp = ops.ConvParam([w], [b])
            cells = []
            for _ in range(a.n):
                x, h = torch.randn(a.batch, Hh, Ww, C, device=dev), torch.randn(a.batch, Hh, Ww, C, device=dev)
                ur = torch.rand(a.batch, Hh, Ww, 2 * C, device=dev)
                g = torch.randn(a.batch, Hh, Ww, cout, device=dev)
                cells.append((x, h, ur, g))
            ws, bws = cp.grad_ws(wino_ok=True)
            assert getattr(ws, "wino6", False)
            kw = dict(x1=None, in_mode=H.IN_CAT, C1=C, dbias=bws) if kind == "ur" else dict(in_mode=H.IN_CAT_MUL, C1=C, dbias=bws, xm_off=C)

            def kwof(c):
                k = dict(kw, x1=c[1])
                if kind == "o":
                    k["xm"] = c[2]
                return k

            def separate():
                for c in cells:
                    ops.wgrad_launch(c[0], taps, c[3], ws, cout, **kwof(c))

            segs = (H.WgradSeg * a.n)()
            for j, c in enumerate(cells):
                segs[j].x0, segs[j].x1, segs[j].dout = ops._p(c[0]), ops._p(c[1]), ops._p(c[3])
                if kind == "o":
                    segs[j].xm = ops._p(c[2], C)

            def merged():
                ops.wgrad_launch(cells[0][0], taps, cells[0][3], ws, cout, segs=segs, **kwof(cells[0]))

            t1, t2 = timeit(separate, a.reps), timeit(merged, a.reps)
            print("gru%d_%-4s %10.3f %10.3f %10.3f" % (i, kind, t1, t2, t2 / t1))
            cp.discard()


if __name__ == "__main__":
    main()
