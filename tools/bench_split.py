#!/usr/bin/env python3
"""Split channel reduction of the latency-bound F(2x2,3x3) launches (csrc/conv_wino.hip): one launch timed at batch 1 for the layer shapes
of the streaming path, unsplit and with 2 / 3 / 4 / 6 / 8 splits forced.  Usage (GPU box): python tools/bench_split.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rpg_ramnet_amd import ops, _hip as H  # noqa: E402
from bench_wino import timeit  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    taps = ops.Taps.get("conv", 3, 1)
    print("%-34s %5s | %s" % ("layer (B = 1)", "WGs", "  ".join("%8s" % s for s in ("unsplit", "auto", "2", "3", "4", "6", "8"))))
    for name, Hh, Ww, cin, cout in (("residual conv 32x43 256->256", 32, 43, 256, 256), ("gru2 gates 32x43 512->512", 32, 43, 512, 512),
                                    ("gru2 candidate 32x43 512->256", 32, 43, 512, 256), ("gru1 gates 64x86 256->256", 64, 86, 256, 256),
                                    ("gru1 candidate 64x86 256->128", 64, 86, 256, 128), ("gru0 candidate 128x172 128->64", 128, 172, 128, 64),
                                    ("encoder-2 view 32x43 512->256", 32, 43, 512, 256), ("encoder-1 view 64x86 256->128", 64, 86, 256, 128)):
        w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=dev) * 0.05)
        b = torch.nn.Parameter(torch.randn(cout, device=dev) * 0.1)
        cp = ops.ConvParam([w], [b])
        x = torch.randn(1, Hh, Ww, cin, device=dev)
        y = torch.empty(1, Hh, Ww, cout, device=dev)
        ops.set_winograd_2x4("off")
        t = []
        for s in (0, 1, 2, 3, 4, 6, 8):
            ops.set_winograd_split(s)
            fn = lambda: ops.conv_launch(x, taps, cp.fwd(), y, cout, bias=cp.bias(), epi=H.EPI_RELU)      # noqa: E731
            timeit(fn, reps=5)                        # (first call of a split count allocates its workspace)
            t.append(1e3 * min(timeit(fn, reps=50) for _ in range(3)))
        ops.set_winograd_split(1)
        ops.set_winograd_2x4("auto")
        tiles = min(-(-Ww // 4) * -(-Hh // 32), -(-Ww // 16) * -(-Hh // 8))
        print("%-34s %5d | %s   us" % (name, tiles * cout // 32, "  ".join("%8.1f" % v for v in t)))
    # the folded decoders (csrc/conv_wino24.hip): the whole layer (padded sum + border GEMM + Winograd launch), only the last one splits
    from rpg_ramnet_amd.model.submodules import UpsampleConvLayer
    for name, Hh, Ww, cin, cout in (("decoder 0 32x43 256->128", 32, 43, 256, 128), ("decoder 1 64x86 128->64", 64, 86, 128, 64)):
        m = UpsampleConvLayer(cin, cout, 5, padding=2).to(dev)
        x, sk = torch.randn(1, Hh, Ww, cin, device=dev), torch.randn(1, Hh, Ww, cin, device=dev)
        t = []
        for s in (0, 1, 2, 3, 4, 6, 8):
            ops.set_winograd_split(s)
            with torch.no_grad():
                fn = lambda: m(x, sk)      # noqa: E731
                timeit(fn, reps=5)
                t.append(1e3 * min(timeit(fn, reps=50) for _ in range(3)))
        ops.set_winograd_split(1)
        print("%-34s %5s | %s   us (whole layer)" % (name, "", "  ".join("%8.1f" % v for v in t)))


if __name__ == "__main__":
    main()
