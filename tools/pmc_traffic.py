#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs per kernel
(mean per dispatch).  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports half of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM):
the corrected read figure doubles it."""
import collections
import csv
import json
import sys


def load(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[(r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-60:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return agg


def main(fetch_csv, write_csv, out_json):
    f, w = load(fetch_csv), load(write_csv)
    out = {}
    for (k, c), v in f.items():
        if c == "FETCH_SIZE":
            out.setdefault(k, {})["fetch_KiB_raw_mean"] = sum(v) / len(v)
            out[k]["dispatches"] = len(v)
    for (k, c), v in w.items():
        if c == "WRITE_SIZE":
            out.setdefault(k, {})["write_KiB_mean"] = sum(v) / len(v)
    for k, d in out.items():
        if "fetch_KiB_raw_mean" in d and "write_KiB_mean" in d:
            d["hbm_bytes_per_launch_corrected"] = (2 * d["fetch_KiB_raw_mean"] + d["write_KiB_mean"]) * 1024
    json.dump(out, open(out_json, "w"), indent=1)
    for k, d in sorted(out.items(), key=lambda kv: -kv[1].get("hbm_bytes_per_launch_corrected", 0))[:12]:
        print(f"{k:60s} {d}")


if __name__ == "__main__":
    main(*sys.argv[1:4])
