#!/usr/bin/env python3
"""HBM-side micro-benchmark of the norm kernels (csrc/norm.hip) at the layer shapes of the bench config:
achieved GB/s of the statistics pass, the apply pass and the backward pair.  Usage (GPU box): python tools/bench_norm.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rpg_ramnet_amd import ops  # noqa: E402


def timeit(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3        # us


def main():
    dev = torch.device("cuda:0")
    print("%-22s %-9s | %9s %8s | %9s %8s | %9s %8s" % ("shape [B,H,W,C]", "kind", "stats us", "GB/s", "fwd us", "GB/s", "bwd us", "GB/s"))
    for B, Hh, W, C in [(8, 256, 344, 32), (8, 128, 172, 64), (8, 64, 86, 128), (8, 32, 43, 256), (8, 256, 344, 1)]:
        for kind in ("BN", "IN"):
            layer = (torch.nn.BatchNorm2d(C) if kind == "BN" else torch.nn.InstanceNorm2d(C, track_running_stats=True)).to(dev).train()
            x = torch.randn(B, Hh, W, C, device=dev, requires_grad=True)
            nbytes = x.numel() * 4
            groups = 1 if kind == "BN" else B
            t_stats = timeit(lambda: ops.norm_stats(x.detach(), groups))
            t_fwd = timeit(lambda: ops.norm_act(x.detach(), layer, "relu"))
            y = ops.norm_act(x, layer, "relu")
            dy = torch.randn_like(y)
            t_bwd = timeit(lambda: torch.autograd.grad(y, x, dy, retain_graph=True))
            # bytes: stats 1 read; forward = stats + (1 read + 1 write); backward = partial (3 reads) + (3 reads + 1 write)
            print("%-22s %-9s | %9.1f %8.0f | %9.1f %8.0f | %9.1f %8.0f" % (
                str([B, Hh, W, C]), kind, t_stats, nbytes / t_stats / 1e3, t_fwd, 3 * nbytes / t_fwd / 1e3, t_bwd, 7 * nbytes / t_bwd / 1e3))


if __name__ == "__main__":
    main()
