#!/usr/bin/env python3
"""Event -> voxel-grid binning at the package-batch size of the bench config: 40 grids x 200 000 events, 5 bins, 256 x 344
(ramnet_voxelize_batch); us per launch and fraction of the HBM roofline on SURVEY 8d's algorithmic bytes (32 B per event read, 2 x 8 B
of vote traffic, one write of the grids).  Usage (GPU box): python tools/bench_voxel.py [grids] [events] [bins] [height] [width]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rpg_ramnet_amd import voxel  # noqa: E402


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
    bins = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    Hh = int(sys.argv[4]) if len(sys.argv) > 4 else 256
    W = int(sys.argv[5]) if len(sys.argv) > 5 else 344
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    ev = np.empty((G, N, 4))
    for g in range(G):
        ev[g] = np.stack([np.sort(rng.uniform(0, 0.05, N)), rng.integers(0, W, N).astype(np.float64), rng.integers(0, Hh, N).astype(np.float64),
                          rng.integers(0, 2, N).astype(np.float64)], 1)
    cat = torch.from_numpy(ev.reshape(G * N, 4)).to(dev)
    off = torch.arange(G + 1, dtype=torch.int64, device=dev) * N
    grids = torch.empty(G, bins, Hh, W, device=dev)
    L = voxel.H.lib()
    st = voxel._st

    def fn():
        voxel.H.check(L.ramnet_voxelize_batch(voxel._p(cat), voxel._p(off), G, N, bins, W, Hh, voxel._p(grids), st()), "ramnet_voxelize_batch")
    fn()
    ref = voxel.events_to_voxel_grid(cat[:N], bins, W, Hh)              # the single-list (global-atomic) entry point
    print("grid 0 vs the single-list kernel: max abs diff %.3e (sum |ref| %.1f)" % (float((grids[0] - ref).abs().max()), float(ref.abs().sum())))
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / reps * 1e3
    nbytes = G * N * (32 + 16) + G * bins * Hh * W * 4
    print("%d grids x %d events: %.1f us per launch, %.1f MB algorithmic -> %.2f TB/s = %.3f of 8 TB/s" % (G, N, us, nbytes / 1e6, nbytes / us / 1e6, nbytes / us / 8e6))


if __name__ == "__main__":
    main()
