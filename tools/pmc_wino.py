#!/usr/bin/env python3
"""Per (kernel, grid, reduction depth) summary of a rocprofv3 --kernel-trace --pmc pass over tools/bench_wino.py: duration, effective
clock, MFMA-busy and the SQ wait buckets as fractions of the wave cycles.  Usage: pmc_wino.py counter_collection.csv kernel_trace.csv"""
import csv
import sys
from collections import defaultdict

dur = {}
for r in csv.DictReader(open(sys.argv[2])):
    dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
per = defaultdict(lambda: defaultdict(float))
order = []
for r in csv.DictReader(open(sys.argv[1])):
    d = r["Dispatch_Id"]
    if d not in per:
        order.append(d)
    per[d]["name"] = r["Kernel_Name"].split("(")[0].replace("void ramnet::", "")
    per[d]["grid"] = r.get("Grid_Size", "")
    per[d][r["Counter_Name"]] += float(r["Counter_Value"])
# consecutive dispatches of the same (kernel, grid) = one timing loop of bench_wino.py (a reduction depth)
groups, last = [], None
for d in order:
    v = per[d]
    if "conv_" not in v["name"]:
        continue
    key = (v["name"], v["grid"])
    if key != last:
        groups.append([key, []])
        last = key
    groups[-1][1].append(d)
print("%-44s %9s %4s %9s %6s %6s | %6s %6s %6s %6s %7s" % ("kernel", "grid", "n", "avg us", "GHz", "MFMA%", "active", "w_inst", "w_any", "w_lds", "ldsconf"))
for key, ds in groups:
    n = len(ds)
    ns = sum(dur.get(d, 0) for d in ds)
    g = lambda c: sum(per[d].get(c, 0.0) for d in ds)      # noqa: E731
    gui = g("GRBM_GUI_ACTIVE") / 8.0
    wc = g("SQ_WAVE_CYCLES") or 1.0
    print("%-44s %9s %4d %9.1f %6.3f %6.1f | %6.3f %6.3f %6.3f %6.3f %7.3f" % (
        key[0][:44], key[1], n, ns / n / 1e3, gui / ns if ns else 0, 100.0 * g("SQ_VALU_MFMA_BUSY_CYCLES") / (gui * 1024) if gui else 0,
        g("SQ_ACTIVE_INST_ANY") / wc, g("SQ_WAIT_INST_ANY") / wc, g("SQ_WAIT_ANY") / wc, g("SQ_WAIT_INST_LDS") / wc,
        g("SQ_LDS_BANK_CONFLICT") / max(1.0, g("SQ_LDS_IDX_ACTIVE"))))
