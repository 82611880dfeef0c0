#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs per kernel
(mean per dispatch).  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports half of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM):
the corrected read figure doubles it."""
import collections
import csv
import json
import sys


def _name(raw):
    """Kernel name without its argument list; the FULL template name is the key (round 4 cut it to its last 60 characters:
    two kernels sharing a tail were merged, and a kernel seen in one pass only got half a figure -- VERDICT r04 item 12)."""
    return raw.replace("(anonymous namespace)::", "").split("(")[0]


def load(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[_name(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return agg


def main(fetch_csv, write_csv, out_json):
    f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    out = {}
    for k in sorted(set(f) | set(w)):
        d = out[k] = {"dispatches_fetch_pass": len(f.get(k, [])), "dispatches_write_pass": len(w.get(k, []))}
        if k in f:
            d["fetch_KiB_raw_mean"] = sum(f[k]) / len(f[k])
        if k in w:
            d["write_KiB_mean"] = sum(w[k]) / len(w[k])
        nf, nw = d["dispatches_fetch_pass"], d["dispatches_write_pass"]
        if nf and nw and abs(nf - nw) <= 0.1 * max(nf, nw):
            d["hbm_bytes_per_launch_corrected"] = (2 * d["fetch_KiB_raw_mean"] + d["write_KiB_mean"]) * 1024
        else:
            # one-sided (the kernel ran in one of the two passes only) or the passes dispatched it a different number of
            # times (another launch mix): no per-launch figure is formed from mismatched halves
            d["hbm_bytes_per_launch_corrected"] = None
            d["note"] = "passes disagree on this kernel's dispatches: no corrected figure"
    json.dump(out, open(out_json, "w"), indent=1)
    for k, d in sorted(out.items(), key=lambda kv: -(kv[1].get("hbm_bytes_per_launch_corrected") or 0))[:12]:
        print(f"{k[-60:]:60s} {d}")


if __name__ == "__main__":
    main(*sys.argv[1:4])
