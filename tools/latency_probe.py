#!/usr/bin/env python3
"""Live batch-1 latency of ONE measurement (update of the state with an event grid, then its decode; BASELINE configs[3]) on an idle GPU:
eager launches with and without the per-scale branch streams (ops.set_branch_overlap), and serial hipGraph replays.
Usage (GPU box): python tools/latency_probe.py [--reps 200]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from rpg_ramnet_amd import ops  # noqa: E402
from rpg_ramnet_amd.graph import GraphedStream, TimeBatchedStream  # noqa: E402
from rpg_ramnet_amd.model.model import ERGB2DepthRecurrent  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--wino2x4", default="auto")
    a = ap.parse_args()
    w24 = a.wino2x4.split(",")
    ops.set_winograd_2x4(w24[0], min_wgs=int(w24[1]) if len(w24) > 1 else None)
    cfg = dict(bench.RELEASED, gpu=0, every_x_rgb_frame=5, baseline=False, loss_composition=["image", "events4"], state_combination="convgru")
    torch.manual_seed(0)
    m = ERGB2DepthRecurrent(cfg)
    m = m.to(m.gpu).eval()
    H, W = 256, 344
    ev = torch.randn(1, 5, H, W, device=m.gpu)

    def timed(fn, sync_each=True):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(a.reps):
            fn()
            if sync_each:
                torch.cuda.synchronize()      # a live sensor waits for the depth map before the next measurement arrives
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t) / a.reps

    st = [m.init_states(1, H, W)]

    def eager():
        with torch.no_grad():
            s, _ = m.update_events(ev, st[0])
            st[0] = s
            return m.decode(s)
    for on in (False, True):
        ops.set_branch_overlap(on)
        print("eager, branch streams %-5s: %.3f ms per update+decode (synchronised after each)" % (on, timed(eager)))
    ops.set_branch_overlap(False)
    for pipelined in (False,):
        g = GraphedStream(m, 1, H, W, pipelined=pipelined)
        print("hipGraph replays (update, decode), serial: %.3f ms" % timed(lambda: g.wait(g.update_events(ev))))
    from rpg_ramnet_amd.graph import LatencyStream
    ls = LatencyStream(m, 1, H, W)
    print("graph.LatencyStream (fine-scale updates on a second stream, 4 replays): %.3f ms" % timed(lambda: ls.update_events(ev)))
    tb = TimeBatchedStream(m, 1, H, W, max_events=1)      # groups of ONE measurement: encoder chain, then the three scales' updates side by side, then the decode
    print("hipGraph chains, per-scale updates side by side (TimeBatchedStream, groups of 1): %.3f ms" % timed(lambda: tb.wait(tb.push_events(ev))))


if __name__ == "__main__":
    main()
