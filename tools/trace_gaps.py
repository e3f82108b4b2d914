#!/usr/bin/env python3
"""GPU idle-gap analysis of a rocprofv3 kernel trace: busy union vs wall, biggest gaps and what preceded them."""
import csv
import sys
from collections import defaultdict

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
rows.sort()
# restrict to the last N launches = steady-state steps
last = rows[-int(sys.argv[2]):] if len(sys.argv) > 2 else rows
t0, t1 = last[0][0], max(e for _, e, _ in last)
busy, cur_end, gaps = 0, last[0][0], []
gap_by = defaultdict(lambda: [0, 0])
prev = None
for s, e, n in last:
    if s > cur_end:
        gaps.append((s - cur_end, prev, n))
        gap_by[(prev, n)][0] += s - cur_end
        gap_by[(prev, n)][1] += 1
        busy += e - s
        cur_end = e
    else:
        if e > cur_end:
            busy += e - cur_end
            cur_end = e
    prev = n
wall = t1 - t0
print("launches %d wall %.1f ms busy %.1f ms (%.1f %%) idle %.1f ms" % (len(last), wall / 1e6, busy / 1e6, 100.0 * busy / wall, (wall - busy) / 1e6))
print("top gap sources (prev kernel -> next kernel): total ms, count, avg us")
for (a, b), (t, c) in sorted(gap_by.items(), key=lambda kv: -kv[1][0])[:25]:
    print("%8.2f %6d %8.1f   %s -> %s" % (t / 1e6, c, t / c / 1e3, a, b))
