#!/usr/bin/env python3
"""Launch-by-launch timeline of ONE live batch-1 measurement (update with an event grid, then its decode; BASELINE configs[3]) from a
rocprofv3 kernel trace: start offset, duration, gap to the previous kernel's end and workgroup count of every launch on the critical path.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -o lt -- python $ROOT/tools/latency_trace.py run [--eager]
    python tools/latency_trace.py table $(find /tmp/lt -name '*kernel_trace.csv')
"""
import csv
import os
import sys
import time

REPS = 12


def run(eager):
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from rpg_ramnet_amd.graph import GraphedStream
    from rpg_ramnet_amd.model.model import ERGB2DepthRecurrent
    cfg = dict(bench.RELEASED, gpu=0, every_x_rgb_frame=5, baseline=False, loss_composition=["image", "events4"], state_combination="convgru")
    torch.manual_seed(0)
    m = ERGB2DepthRecurrent(cfg)
    m = m.to(m.gpu).eval()
    H, W = 256, 344
    ev = torch.randn(1, 5, H, W, device=m.gpu)
    if "--latency-stream" in sys.argv:
        from rpg_ramnet_amd.graph import LatencyStream
        ls = LatencyStream(m, 1, H, W)

        def one():
            ls.update_events(ev)
    elif eager:
        st = [m.init_states(1, H, W)]

        def one():
            with torch.no_grad():
                st[0], _ = m.update_events(ev, st[0])
                return m.decode(st[0])
    else:
        g = GraphedStream(m, 1, H, W, pipelined=False)

        def one():
            g.wait(g.update_events(ev))
    for _ in range(REPS):
        one()
        torch.cuda.synchronize()
        time.sleep(0.002)            # idle gap: separates the measurements in the trace


def table(path):
    rows = [r for r in csv.DictReader(open(path))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # measurements = runs of launches separated by >= 1 ms of idle time
    groups, cur, last_end = [], [], None
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if last_end is not None and s - last_end > 1_000_000 and cur:
            groups.append(cur)
            cur = []
        cur.append(r)
        last_end = max(last_end or 0, e)
    if cur:
        groups.append(cur)
    n = len(groups[-1])
    same = [g for g in groups if len(g) == n]
    g = same[-1]
    t0 = int(g[0]["Start_Timestamp"])
    print("%d measurements of %d launches in the trace; the last one:" % (len(same), n))
    print("%9s %8s %8s %6s %5s  %s" % ("start us", "dur us", "gap us", "WGs", "queue", "kernel"))
    prev_end, busy = t0, 0.0
    for r in g:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        wg = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1)) * int(r.get("Grid_Size_Z", 1)) // max(
            1, int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1)) * int(r.get("Workgroup_Size_Z", 1)))
        name = r["Kernel_Name"].replace("ramnet::", "").replace("void ", "").split("(")[0]
        print("%9.1f %8.1f %8.1f %6d %5s  %s" % ((s - t0) * 1e-3, (e - s) * 1e-3, (s - prev_end) * 1e-3, wg, r.get("Queue_Id", "")[-3:], name[:90]))
        busy += (e - s) * 1e-3
        prev_end = max(prev_end, e)
    span = [(int(x[-1]["End_Timestamp"]) - int(x[0]["Start_Timestamp"])) * 1e-3 for x in same[2:]]
    print("first start -> last end: %.1f us (this one), mean %.1f us over %d; kernel time %.1f us, gaps %.1f us" % (
        (prev_end - t0) * 1e-3, sum(span) / max(1, len(span)), len(span), busy, (prev_end - t0) * 1e-3 - busy))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run("--eager" in sys.argv)
    else:
        table(sys.argv[2])
