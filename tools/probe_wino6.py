#!/usr/bin/env python3
"""Where does a workgroup of conv_wino_r6_kernel spend its life?  Needs the probe build (tools/probe_wino6.sh):
    RAMNET_HIP_LIB=rpg_ramnet_amd/abl/lib_probe6.so python tools/probe_wino6.py
Thread 0 of every workgroup stamps the shader clock at: 0 entry, 1 descriptors / prefetcher set up, 2 first patch in LDS (first barrier),
3 prologue done (first transforms, second barrier), 4 main loop done, 5 exchange written (barrier), 6 epilogue stores retired; plus the 100 MHz
wall counter at entry and exit and the CU the workgroup ran on.  ConvGRU gate / candidate shapes of the training step (B = 8), F(2x4) forced."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rpg_ramnet_amd import ops, _hip as H  # noqa: E402

NWG, NS = 16384, 16
NAMES = ["setup", "first patch -> LDS", "first transforms", "main loop", "exchange", "epilogue"]


def read(which="ramnet_probe6_read"):
    buf = np.zeros(NWG * NS, dtype=np.uint64)
    fn = getattr(H.lib(), which)
    fn.argtypes, fn.restype = [C.c_void_p, C.c_size_t], C.c_int
    assert fn(buf.ctypes.data, buf.size) == 0
    return buf.reshape(NWG, NS).astype(np.int64)


WNAMES = ["setup", "first strips -> LDS", "first transforms", "main loop", "slab join", "bias"]


def wgrad():
    """conv_wgrad_wino_r6_kernel on the six ConvGRU backward-weights launches of one cell update (B = 8): stamps 0 entry, 1 set up, 2 first strips
    in LDS, 3 first transforms, 4 main loop done, 5 slab join retired, 6 bias partials done."""
    dev = torch.device("cuda:0")
    ops.set_wgrad_winograd_2x4("force")
    taps = ops.Taps.get("conv", 3, 1)
    B = 8
    for i, Cc in enumerate([64, 128, 256]):
        Hh, Ww = 256 >> (i + 1), 344 >> (i + 1)
        for kind, cout in (("gates", 2 * Cc), ("candidate", Cc)):
            w = torch.nn.Parameter(torch.randn(cout, 2 * Cc, 3, 3, device=dev) * 0.01)
            b = torch.nn.Parameter(torch.zeros(cout, device=dev))
            cp = ops.ConvParam([w], [b])
            x, h = torch.randn(B, Hh, Ww, Cc, device=dev), torch.randn(B, Hh, Ww, Cc, device=dev)
            ur, g = torch.rand(B, Hh, Ww, 2 * Cc, device=dev), torch.randn(B, Hh, Ww, cout, device=dev)
            ws, bws = cp.grad_ws(wino_ok=True)
            kw = dict(x1=h, in_mode=H.IN_CAT, C1=Cc, dbias=bws) if kind == "gates" else dict(x1=h, in_mode=H.IN_CAT_MUL, C1=Cc, dbias=bws, xm=ur, xm_off=Cc)

            def launch():
                ops.wgrad_launch(x, taps, g, ws, cout, **kw)
            for _ in range(3):
                launch()
            torch.cuda.synchronize()
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record()
            for _ in range(10):
                launch()
            e_.record()
            torch.cuda.synchronize()
            t_ms = s_.elapsed_time(e_) / 10
            before = read("ramnet_probe_w6_read")
            launch()
            torch.cuda.synchronize()
            t = read("ramnet_probe_w6_read")
            full = (t[:, 0] != before[:, 0]) & (t[:, 6] != before[:, 6])
            t = t[full]
            n = int(full.sum())
            d = np.diff(t[:, :7], axis=1).astype(np.float64)
            life = (t[:, 6] - t[:, 0]).astype(np.float64)
            wall = (t[:, 10] - t[:, 7]) * 10.0
            ghz = float(np.median(life / np.maximum(wall, 1.0)))
            start, end = (t[:, 7] - t[:, 7].min()) * 0.01, (t[:, 10] - t[:, 7].min()) * 0.01
            print("\ngru%d %s (%dx%d, Cin %d -> Cout %d): %.1f us per launch (events), %d workgroups, shader clock %.2f GHz; starts %.1f .. %.1f us, "
                  "last end %.1f us" % (i, kind, Hh, Ww, 2 * Cc, cout, t_ms * 1e3, n, ghz, start.min(), start.max(), end.max()))
            print("  life of a workgroup: median %.1f us (p10 %.1f, p90 %.1f): " % (np.median(life) / ghz * 1e-3, np.percentile(life, 10) / ghz * 1e-3,
                                                                                  np.percentile(life, 90) / ghz * 1e-3) +
                  "  ".join("%s %.2f us" % (WNAMES[k], np.median(d[:, k]) / ghz * 1e-3) for k in range(6)))
            ts = np.linspace(0, end.max(), 21)[1:-1]
            print("  workgroups alive at 5 %% steps of the launch: %s" % " ".join(str(int(((start <= x_) & (end > x_)).sum())) for x_ in ts))
            cp.discard()


def split():
    """conv_wino_r6s_kernel (split bf16 operands, ONE workgroup per CU) on the six forward launches of a ConvGRU update (B = 8): stamps 0 entry, 1 set up,
    2 first patch in LDS, 3 prologue done, 4 main loop done, 5 / 11 exchange of the first / second 32-channel half written, 6 / 12 its epilogue's stores
    issued, 13 stores retired."""
    dev = torch.device("cuda:0")
    taps = ops.Taps.get("conv", 3, 1)
    ops.set_winograd_2x4("force")
    ops.set_split_operands(True)
    B = 8
    names = ["set-up", "first patch -> LDS", "row + first A operand", "main loop", "exchange 0", "epilogue 0", "exchange 1", "epilogue 1", "stores retired"]
    for i, Cc in enumerate([64, 128, 256]):
        Hh, Ww = 256 >> (i + 1), 344 >> (i + 1)
        for kind, cout in (("gates", 2 * Cc), ("candidate", Cc)):
            w = torch.nn.Parameter(torch.randn(cout, 2 * Cc, 3, 3, device=dev) * 0.02)
            b = torch.nn.Parameter(torch.randn(cout, device=dev) * 0.1)
            cp = ops.ConvParam([w], [b])
            x, h = torch.randn(B, Hh, Ww, Cc, device=dev), torch.tanh(torch.randn(B, Hh, Ww, Cc, device=dev))
            ur, hr, hn, o = (torch.rand(B, Hh, Ww, n, device=dev) for n in (2 * Cc, Cc, Cc, Cc))
            if kind == "gates":
                def launch():
                    ops.conv_launch(x, taps, cp.fwd(), ur, 2 * Cc, x1=h, in_mode=H.IN_CAT, C1=Cc, bias=cp.bias(), epi=H.EPI_SIGMOID_HR, e1=h, o1=hr)
            else:
                def launch():
                    ops.conv_launch(x, taps, cp.fwd(), hn, Cc, x1=hr, in_mode=H.IN_CAT, C1=Cc, bias=cp.bias(), epi=H.EPI_GRU_BLEND, e0=ur, e1=h, o1=o)
            for _ in range(3):
                launch()
            torch.cuda.synchronize()
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record()
            for _ in range(10):
                launch()
            e_.record()
            torch.cuda.synchronize()
            t_ms = s_.elapsed_time(e_) / 10
            before = read("ramnet_probe6s_read")
            launch()
            torch.cuda.synchronize()
            t = read("ramnet_probe6s_read")
            full = (t[:, 0] != before[:, 0]) & (t[:, 13] != before[:, 13])
            t = t[full]
            n, nch = int(full.sum()), 2 * Cc // 16
            seq = [0, 1, 2, 3, 4, 5, 6, 11, 12, 13]
            d = np.diff(t[:, seq], axis=1).astype(np.float64)
            life = (t[:, 13] - t[:, 0]).astype(np.float64)
            wall = (t[:, 10] - t[:, 7]) * 10.0
            ghz = float(np.median(life / np.maximum(wall, 1.0)))
            start, end = (t[:, 7] - t[:, 7].min()) * 0.01, (t[:, 10] - t[:, 7].min()) * 0.01
            print("\ngru%d %s (%dx%d, Cin %d -> Cout %d): %.1f us per launch (events; %s), %d workgroups, shader clock %.2f GHz; starts %.1f .. %.1f us, "
                  "last end %.1f us" % (i, kind, Hh, Ww, 2 * Cc, cout, t_ms * 1e3, H.lib().ramnet_last_kernel().decode(), n, ghz, start.min(), start.max(), end.max()))
            print("  life of a workgroup: median %.1f us (p10 %.1f, p90 %.1f):  " % (np.median(life) / ghz * 1e-3, np.percentile(life, 10) / ghz * 1e-3,
                                                                                   np.percentile(life, 90) / ghz * 1e-3) +
                  "  ".join("%s %.2f" % (names[k], np.median(d[:, k]) / ghz * 1e-3) for k in range(9)))
            ml = np.median(d[:, 3])
            print("  main loop: %.0f cycles per 16-channel chunk (%d chunks; 72 MFMAs = 2304 cycles of the pipe); fixed part %.2f us of %.2f" % (
                ml / nch, nch, (np.median(life) - ml) / ghz * 1e-3, np.median(life) / ghz * 1e-3))
            ts = np.linspace(0, end.max(), 21)[1:-1]
            print("  workgroups alive at 5 %% steps of the launch: %s" % " ".join(str(int(((start <= x_) & (end > x_)).sum())) for x_ in ts))


def main():
    if "--wgrad" in sys.argv:
        return wgrad()
    if "--split" in sys.argv:
        return split()
    dev = torch.device("cuda:0")
    taps = ops.Taps.get("conv", 3, 1)
    ops.set_winograd(True)
    ops.set_winograd_2x4("force")
    B = 8
    for (Hh, Ww, cin, cout, epi) in ((128, 172, 128, 128, H.EPI_SIGMOID), (128, 172, 128, 64, H.EPI_LINEAR), (64, 86, 256, 256, H.EPI_SIGMOID),
                                      (32, 43, 512, 512, H.EPI_SIGMOID), (128, 172, 64, 64, H.EPI_RELU), (128, 172, 32, 64, H.EPI_RELU)):
        w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=dev) * 0.05)
        b = torch.nn.Parameter(torch.randn(cout, device=dev) * 0.1)
        cp = ops.ConvParam([w], [b])
        x = torch.randn(B, Hh, Ww, cin, device=dev)
        y = torch.empty(B, Hh, Ww, cout, device=dev)

        def launch():
            ops.conv_launch(x, taps, cp.fwd(), y, cout, bias=cp.bias(), epi=epi)
        for _ in range(3):
            launch()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            launch()
        e.record()
        torch.cuda.synchronize()
        t_ms = s.elapsed_time(e) / 10
        before = read()
        launch()
        torch.cuda.synchronize()
        t = read()
        ran = t[:, 0] != before[:, 0]
        full = ran & (t[:, 6] != before[:, 6])
        t = t[full]
        n, nch = int(full.sum()), cin // 8
        d = np.diff(t[:, :7], axis=1).astype(np.float64)              # cycles per phase
        life = (t[:, 6] - t[:, 0]).astype(np.float64)
        wall = (t[:, 10] - t[:, 7]) * 10.0                            # ns
        ghz = float(np.median(life / np.maximum(wall, 1.0)))
        start = (t[:, 7] - t[:, 7].min()) * 0.01                      # us
        end = (t[:, 10] - t[:, 7].min()) * 0.01
        cu = (t[:, 9] & 0xf) * 1000 + ((t[:, 8] >> 13) & 7) * 100 + ((t[:, 8] >> 8) & 0xf)      # (xcc, se, cu)
        print("\n%dx%d Cin %d -> Cout %d, B %d: %.1f us per launch (events), %d workgroups (%d launched) on %d CUs, shader clock %.2f GHz" % (
            Hh, Ww, cin, cout, B, t_ms * 1e3, n, int(ran.sum()), len(np.unique(cu)), ghz))
        print("  workgroup starts %.1f .. %.1f us, last end %.1f us; life of a workgroup: median %.1f us (%.0f cycles), p10 %.1f, p90 %.1f" % (
            start.min(), start.max(), end.max(), np.median(life) / ghz * 1e-3, np.median(life), np.percentile(life, 10) / ghz * 1e-3,
            np.percentile(life, 90) / ghz * 1e-3))
        order = np.argsort(start)
        first, rest = order[:min(512, n)], order[min(512, n):]
        for label, idx in (("all", order), ("the first 512 started", first), ("the later ones", rest)):
            if len(idx) == 0:
                continue
            print("  %-22s" % label + "  ".join("%s %.2f us" % (NAMES[k], np.median(d[idx, k]) / ghz * 1e-3) for k in range(6)))
        e5 = t[:, 5].astype(np.float64)
        ep = [np.median(t[:, k] - (e5 if k == 11 else t[:, k - 1])) / ghz * 1e-3 for k in (11, 12, 13, 14)] + [np.median(t[:, 6] - t[:, 14]) / ghz * 1e-3]
        print("  epilogue: operands of half 0 requested %.2f us, its 4 quads out of LDS / activated / stores issued %.2f, half 1: %.2f + %.2f, "
              "stores retired %.2f" % tuple(ep))
        ml = np.median(d[:, 3])
        print("  main loop: %.0f cycles per chunk (%d chunks; 24 MFMAs = 1536 cycles of the pipe per wave, two waves per SIMD); fixed part of a "
              "workgroup's life: %.2f us of %.2f" % (ml / nch, nch, (np.median(life) - ml) / ghz * 1e-3, np.median(life) / ghz * 1e-3))
        # occupancy over time: how many workgroups are alive at 20 points of the launch
        ts = np.linspace(0, end.max(), 21)[1:-1]
        alive = [(int(((start <= x) & (end > x)).sum())) for x in ts]
        print("  workgroups alive at 5 %% steps of the launch: %s" % " ".join(str(a) for a in alive))


if __name__ == "__main__":
    main()
