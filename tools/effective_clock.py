#!/usr/bin/env python3
"""Effective shader clock under load = GRBM_GUI_ACTIVE (per XCD) / kernel wall time, and MFMA-busy, per kernel, from a
`rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES` run (MI355X_MICROARCH.md, DVFS give-back: the chip
clocks to its power budget, so the 2.4 GHz peak of the roofline is not what a sustained MFMA kernel sees).
Usage: python tools/effective_clock.py counter_collection.csv [kernel_trace.csv]"""
import csv
import sys
from collections import defaultdict

dur = {}
if len(sys.argv) > 2:
    for r in csv.DictReader(open(sys.argv[2])):
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
per = defaultdict(lambda: defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    d = r["Dispatch_Id"]
    per[d]["name"] = r["Kernel_Name"].split("(")[0].replace("void ramnet::", "")[:40]
    per[d]["grid"] = r.get("Grid_Size", "")
    per[d][r["Counter_Name"]] += float(r["Counter_Value"])
    if "Start_Timestamp" in r and r["Start_Timestamp"]:
        per[d]["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    elif d in dur:
        per[d]["ns"] = dur[d]
agg = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for d, v in per.items():
    if "GRBM_GUI_ACTIVE" not in v or not v.get("ns"):
        continue
    a = agg[(v["name"], v["grid"])]
    a[0] += 1
    a[1] += v["ns"]
    a[2] += v["GRBM_GUI_ACTIVE"] / 8.0
    a[3] += v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
print("%-42s %9s %6s %10s %8s %9s" % ("kernel", "grid", "calls", "avg us", "GHz", "MFMA busy"))
for (n, g), (c, ns, cyc, mf) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if ns / c < 20e3:
        continue
    print("%-42s %9s %6d %10.1f %8.3f %8.1f%%" % (n, g, c, ns / c / 1e3, cyc / ns, 100.0 * mf / (cyc * 1024) if cyc else 0.0))
