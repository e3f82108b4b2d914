#!/usr/bin/env python3
"""Per-kernel table of the batch-1 streaming chain from a rocprofv3 kernel trace (tools/profile_stream.sh): for every kernel
symbol (template arguments kept) and grid size: launches, average / min duration, workgroups — the launch-by-launch view of one
update + decode that the whole-step number hides."""
import csv
import sys
from collections import defaultdict


def main(path):
    rows = list(csv.DictReader(open(path)))
    agg = defaultdict(list)
    for r in rows:
        name = r.get("Kernel_Name", "")
        if "ramnet" not in name:
            continue
        name = name.replace("ramnet::", "").split("(")[0]
        wg = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1)) * int(r.get("Grid_Size_Z", 1)) // max(1, int(r["Workgroup_Size_X"]))
        agg[(name, wg)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    tot = sum(sum(v) for v in agg.values())
    print("%-64s %8s %8s %9s %9s %7s" % ("kernel", "WGs", "launches", "avg us", "min us", "share"))
    for (name, wg), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("%-64s %8d %8d %9.1f %9.1f %6.1f%%" % (name[:64], wg, len(v), sum(v) / len(v), min(v), 100 * sum(v) / tot))
    print("total kernel time %.1f ms over %d launches" % (tot * 1e-3, sum(len(v) for v in agg.values())))


if __name__ == "__main__":
    main(sys.argv[1])
