#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same command.
Input: the two counter_collection.csv files; output: JSON {kernel: {grid: {fetch_kb_raw, write_kb, hbm_bytes_per_launch, launches}}}.
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE (KB) counts 1/2 of wide coalesced reads -> doubled here.
Usage: python tools/pmc_traffic.py fetch.csv write.csv out.json"""
import csv
import json
import re
import sys
from collections import defaultdict


def norm(name):
    name = name.split("(")[0].replace("void ", "").replace("ramnet::", "")
    name = re.sub(r"\s+", "", name)
    return name.replace("false", "0").replace("true", "1")


def collect(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r.get("Kernel_Name") or r.get("Kernel Name")
        if "ramnet" not in k:
            continue
        acc[(norm(k), r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
    return acc


def main(fetch_csv, write_csv, out_json):
    f, w = collect(fetch_csv, "FETCH_SIZE"), collect(write_csv, "WRITE_SIZE")
    out = defaultdict(dict)
    for key in sorted(set(f) | set(w)):
        fk = sum(f.get(key, [0])) / max(1, len(f.get(key, [])))
        wk = sum(w.get(key, [0])) / max(1, len(w.get(key, [])))
        out[key[0]][key[1]] = {"fetch_kb_raw": fk, "write_kb": wk, "hbm_bytes_per_launch": (2 * fk + wk) * 1024,
                               "launches": max(len(f.get(key, [])), len(w.get(key, [])))}
    json.dump(out, open(out_json, "w"), indent=1)
    for k, v in out.items():
        n = sum(e["launches"] for e in v.values())
        print("%-44s %6d launches  %8.1f MB/launch (launch-weighted)" %
              (k, n, sum(e["hbm_bytes_per_launch"] * e["launches"] for e in v.values()) / max(1, n) / 1e6))


if __name__ == "__main__":
    main(*sys.argv[1:4])
