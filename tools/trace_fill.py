#!/usr/bin/env python3
"""How full is the chip over a training step?  Sweep over a rocprofv3 kernel trace (--kernel-trace, csv): at every instant sum
the workgroup slots the running kernels could occupy (grid / workgroup size, 2 slots per CU for <= 256-thread workgroups, 1 for
512) and report the time-weighted distribution of min(1, demand / capacity), the idle time, and which kernels run while the
demand is low.  Usage: python tools/trace_fill.py kernel_trace.csv [last N launches]"""
import csv
import sys
from collections import defaultdict

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    wg = max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
    blocks = max(1, int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // wg)
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("ramnet::", "").replace("void ", "")[:44], blocks * (2 if wg > 256 else 1)))
rows.sort()
if len(sys.argv) > 2:
    rows = rows[-int(sys.argv[2]):]
ev = []
for i, (s, e, n, d) in enumerate(rows):
    ev.append((s, 1, i))
    ev.append((e, 0, i))
ev.sort()
CAP = 512.0
live = set()
hist = defaultdict(float)
low_by = defaultdict(float)
alone = defaultdict(float)
prev_t = ev[0][0]
for t, kind, i in ev:
    dt = t - prev_t
    if dt > 0:
        dem = sum(rows[j][3] for j in live)
        fill = min(1.0, dem / CAP)
        hist["idle" if not live else "<25%" if fill < 0.25 else "<50%" if fill < 0.5 else "<75%" if fill < 0.75 else "<100%" if fill < 1 else "full"] += dt
        if live and fill < 0.75:
            for j in live:
                low_by[rows[j][2]] += dt
        if len(live) == 1:
            alone[rows[next(iter(live))][2]] += dt
    prev_t = t
    if kind:
        live.add(i)
    else:
        live.discard(i)
wall = ev[-1][0] - ev[0][0]
print("launches %d, wall %.1f ms" % (len(rows), wall / 1e6))
for k in ("idle", "<25%", "<50%", "<75%", "<100%", "full"):
    print("  demand %-6s %8.2f ms  %5.1f %%" % (k, hist[k] / 1e6, 100.0 * hist[k] / wall))
print("kernels present while demand < 75 % (ms):")
for n, t in sorted(low_by.items(), key=lambda kv: -kv[1])[:14]:
    print("  %8.2f  %s" % (t / 1e6, n))
print("kernels running ALONE (ms):")
for n, t in sorted(alone.items(), key=lambda kv: -kv[1])[:14]:
    print("  %8.2f  %s" % (t / 1e6, n))
