#!/usr/bin/env python3
"""Per-layer kernel micro-benchmark (forward / backward-data / backward-weights) at the bench shapes.
Usage (GPU box): python tools/bench_layers.py [--batch 8] [--reps 5] [--only gru]"""
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
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=344)
    ap.add_argument("--only", default="")
    ap.add_argument("--wgrad-wino-nf", type=int, default=0)
    ap.add_argument("--wgrad-2x4", type=int, default=1)
    ap.add_argument("--wino2x4", default="auto", help="auto | off | force[,min_wgs]: F(2x4,3x3) selection (ops.set_winograd_2x4)")
    a = ap.parse_args()
    if a.wgrad_wino_nf:
        from rpg_ramnet_amd import _hip as H_
        H_.check(H_.lib().ramnet_set_option(b"wgrad_wino_nf", a.wgrad_wino_nf), "set_option")
    w24 = a.wino2x4.split(",")
    ops.set_winograd_2x4(w24[0], min_wgs=int(w24[1]) if len(w24) > 1 else None)
    dev = torch.device("cuda:0")
    B, Hh, Ww = a.batch, a.height, a.width
    # (name, kind, Cin(real), Cout, k, stride, Hin, Win)   kind: conv | up | gru_ur | gru_o | lstm
    L = []
    L.append(("head_events", "conv", 5, 32, 5, 1, Hh, Ww))
    for i, (ci, co) in enumerate([(32, 64), (64, 128), (128, 256)]):
        L.append(("enc%d" % i, "conv", ci, co, 5, 2, Hh >> i, Ww >> i))
    for i, c in enumerate([64, 128, 256]):
        L.append(("gru%d_ur" % i, "gru_ur", 2 * c, 2 * c, 3, 1, Hh >> (i + 1), Ww >> (i + 1)))
        L.append(("gru%d_o" % i, "gru_o", 2 * c, c, 3, 1, Hh >> (i + 1), Ww >> (i + 1)))
    L.append(("res_conv", "conv", 256, 256, 3, 1, Hh >> 3, Ww >> 3))
    for i, (ci, co) in enumerate([(256, 128), (128, 64), (64, 32)]):
        L.append(("dec%d" % i, "up", ci, co, 5, 1, Hh >> (2 - i), Ww >> (2 - i)))
    for i, c in enumerate([64, 128, 256]):
        L.append(("lstm%d" % i, "lstm", 2 * c, 4 * c, 3, 1, Hh >> (i + 1), Ww >> (i + 1)))
    # the decoder convolutions on a MATERIALISED upsampled input (plain loader): what the fused bilinear loader costs
    for i, (ci, co) in enumerate([(256, 128), (128, 64), (64, 32)]):
        L.append(("dec%d_plain" % i, "conv", ci, co, 5, 1, Hh >> (2 - i), Ww >> (2 - i)))
    print("%-12s %8s | %8s %7s | %8s %7s | %8s %7s" % ("layer", "GFLOP", "fwd ms", "TF/s", "dgrad ms", "TF/s", "wgrad ms", "TF/s"))
    tot = [0.0, 0.0, 0.0, 0.0]
    for name, kind, cin, cout, k, stride, Hin, Win in L:
        if a.only and a.only not in name:
            continue
        pad = k // 2
        cin_p = (cin + 3) // 4 * 4
        Ho, Wo = (Hin + 2 * pad - k) // stride + 1, (Win + 2 * pad - k) // stride + 1
        w = torch.randn(cout, cin, k, k, device=dev) * 0.05
        b = torch.randn(cout, device=dev) * 0.1
        wp = torch.nn.Parameter(w)
        bp = torch.nn.Parameter(b)
        cp = ops.ConvParam([wp], [bp], gates=4 if kind == "lstm" else 1)
        taps, tapsd = ops.Taps.get("conv", k, pad), ops.Taps.get("dgrad1", k, pad)
        gfl = 2.0 * B * Ho * Wo * k * k * cin * cout / 1e9
        ws = torch.zeros((24 if k == 3 else k * k) * cin_p * cout, device=dev)
        ws.wino = k == 3 and stride == 1 and kind != "up" and ops.get_winograd()      # Winograd backward-weights workspace
        if ws.wino and a.wgrad_2x4 and kind != "s2d" and cin_p >= 64:
            ws.wino, ws.wino6 = False, True                                            # F(2x4,3x3) backward-weights
        ws.head_cin = cin if (kind == "conv" and k == 5 and stride == 1 and H.lib().ramnet_head_supported(cin, cout)) else 0
        bws = torch.zeros(cout, device=dev)
        if kind == "conv":
            x = torch.randn(B, Hin, Win, cin_p, device=dev)
            y = torch.empty(B, Ho, Wo, cout, device=dev)
            dy = torch.randn_like(y)
            dx = torch.empty(B, Hin, Win, cin_p, device=dev)
            f = lambda: ops.conv_launch(x, taps, cp.fwd(), y, cout, stride=stride, bias=b, epi=H.EPI_RELU)  # noqa: E731
            if stride == 1:
                d = lambda: ops.conv_launch(dy, tapsd, cp.bwd(), dx, cin, xm=y, in_mode=H.IN_RELUMASK)  # noqa: E731
            else:
                def d():
                    ops.conv_launch_multi(dy, cp.bwd(), dx, cin,
                                          [(ops.Taps.get("dgrad2", k, pad, py, px), (Hin - py + 1) // 2, (Win - px + 1) // 2, (2, 2, py, px))
                                           for py in range(2) for px in range(2)], xm=y, in_mode=H.IN_RELUMASK)
            g = lambda: ops.wgrad_launch(x, taps, dy, ws, cout, stride=stride, gmask=y, dbias=bws)  # noqa: E731
        elif kind == "up":
            x = torch.randn(B, Hin // 2, Win // 2, cin, device=dev)
            s = torch.randn_like(x)
            y = torch.empty(B, Hin, Win, cout, device=dev)
            dy = torch.randn_like(y)
            dx = torch.empty(B, Hin, Win, cin, device=dev)
            if ops.get_fold_upsample():     # four 4x4 parity convolutions of the low-res sum + border bands (ops._folded_upsample_conv)
                f = lambda: ops._folded_upsample_conv(x, s, cp, y, H.EPI_RELU)  # noqa: E731
            else:
                f = lambda: ops.conv_launch(x, taps, cp.fwd(), y, cout, x1=s, in_mode=H.IN_UP2X_SKIP, Hin=Hin, Win=Win, bias=b, epi=H.EPI_RELU)  # noqa: E731
            d = lambda: ops.conv_launch(dy, tapsd, cp.bwd(), dx, cin, xm=y, in_mode=H.IN_RELUMASK)  # noqa: E731
            if ops.get_fold_upsample() and ops._fold_dgrad_ok(B, Hin, Win, cp):     # adjoint of the folded operator (incl. un-pad fold, border adjoint)
                d = lambda: ops._folded_upsample_dgrad(x, dy, y, cp)  # noqa: E731
            if ops.get_fold_upsample():
                g = lambda: ops._folded_upsample_wgrad(x, s, dy, y, cp)  # noqa: E731
            else:
                g = lambda: ops.wgrad_launch(x, taps, dy, ws, cout, x1=s, in_mode=H.IN_UP2X_SKIP, Hin=Hin, Win=Win, gmask=y, dbias=bws)  # noqa: E731
        else:
            C = cin // 2
            x = torch.randn(B, Hin, Win, C, device=dev)
            h = torch.randn_like(x)
            ur = torch.rand(B, Hin, Win, 2 * C, device=dev)
            dx = torch.empty(B, Hin, Win, 2 * C, device=dev)
            if kind == "gru_ur":
                y = torch.empty(B, Hin, Win, 2 * C, device=dev)
                f = lambda: ops.conv_launch(x, taps, cp.fwd(), y, 2 * C, x1=h, in_mode=H.IN_CAT, C1=C, bias=b, epi=H.EPI_SIGMOID)  # noqa: E731
                dy = torch.randn_like(y)
                g = lambda: ops.wgrad_launch(x, taps, dy, ws, 2 * C, x1=h, in_mode=H.IN_CAT, C1=C, dbias=bws)  # noqa: E731
            elif kind == "gru_o":
                y = torch.empty(B, Hin, Win, C, device=dev)
                o = torch.empty_like(y)
                f = lambda: ops.conv_launch(x, taps, cp.fwd(), y, C, x1=h, xm=ur, xm_off=C, in_mode=H.IN_CAT_MUL, C1=C, bias=b,  # noqa: E731
                                            epi=H.EPI_GRU_BLEND, e0=ur, e1=h, o1=o)
                dy = torch.randn_like(y)
                g = lambda: ops.wgrad_launch(x, taps, dy, ws, C, x1=h, xm=ur, xm_off=C, in_mode=H.IN_CAT_MUL, C1=C, dbias=bws)  # noqa: E731
            else:
                y = torch.empty(B, Hin, Win, C, device=dev)
                cn, gates = torch.empty_like(y), torch.empty(B, Hin, Win, 4 * C, device=dev)
                f = lambda: ops.conv_launch(x, taps, cp.fwd(), y, C, x1=h, in_mode=H.IN_CAT, C1=C, bias=b, epi=H.EPI_LSTM, e1=h, o1=cn, o2=gates)  # noqa: E731
                dy = torch.randn(B, Hin, Win, 4 * C, device=dev)
                g = lambda: ops.wgrad_launch(x, taps, dy, ws, 4 * C, x1=h, in_mode=H.IN_CAT, C1=C, dbias=bws)  # noqa: E731
            d = lambda: ops.conv_launch(dy, tapsd, cp.bwd(), dx, 2 * C)  # noqa: E731
        tf = timeit(f, a.reps)
        kf = H.lib().ramnet_last_kernel().decode()
        td, tg = timeit(d, a.reps), timeit(g, a.reps)
        print("%-12s %8.2f | %8.3f %7.1f | %8.3f %7.1f | %8.3f %7.1f  %s" % (name, gfl, tf, gfl / tf, td, gfl / td, tg, gfl / tg, kf))
        if kind != "lstm":
            tot[0] += gfl
            tot[1] += tf
            tot[2] += td
            tot[3] += tg
    print("%-12s %8.2f | %8.3f %7.1f | %8.3f %7.1f | %8.3f %7.1f  (GRU config, one pass; res_conv counted once of 4)"
          % ("sum", tot[0], tot[1], tot[0] / tot[1], tot[2], tot[0] / tot[2], tot[3], tot[0] / tot[3]))


if __name__ == "__main__":
    main()
