#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel name (averages per dispatch)."""
import csv
import sys
from collections import defaultdict


def main(path):
    per = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = r.get("Kernel_Name") or r.get("Kernel Name")
        if "ramnet" not in name:
            continue
        key = (name.split("(")[0].replace("void ramnet::", "")[:44], r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""))
        per[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    ctrs = sorted({c for v in per.values() for c in v})
    print("%-46s %9s %7s | " % ("kernel", "grid", "lds") + " ".join("%14s" % c.replace("SQ_", "")[:14] for c in ctrs))
    for key, v in sorted(per.items()):
        print("%-46s %9s %7s | " % key + " ".join("%14.4g" % (sum(v[c]) / max(1, len(v[c]))) for c in ctrs))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles are summed over all 1024 SIMDs
            gui = sum(v["GRBM_GUI_ACTIVE"]) / 8.0
            mf = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"]) / (gui * 256 * 4)
            print("%-46s   MFMA pipe busy = %.1f %% of SIMD-cycles (GUI_ACTIVE/8 x 1024 SIMDs)" % ("", 100 * mf))


if __name__ == "__main__":
    main(sys.argv[1])
