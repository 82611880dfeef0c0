#!/usr/bin/env python3
"""Mean per-dispatch value of every counter in rocprofv3 --pmc counter_collection CSVs, for the
kernels whose name contains the given substring:  pmc_sq.py <substring> file.csv [file.csv ...]"""
import collections
import csv
import sys


def main(sub, *paths):
    agg = collections.defaultdict(list)
    for p in paths:
        for r in csv.DictReader(open(p)):
            if sub in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in sorted(agg.items()):
        print(f"{c:32s} n={len(v):4d} mean={sum(v) / len(v):16.1f}")


if __name__ == "__main__":
    main(*sys.argv[1:])
