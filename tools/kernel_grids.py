#!/usr/bin/env python3
"""Grid sizes per kernel name from a rocprofv3 --kernel-trace --output-format csv dump:
   kernel_grids.py <kernel_trace.csv> [substring]  ->  name, grid, workgroup, calls, average us"""
import collections
import csv
import sys

rows = collections.defaultdict(lambda: [0, 0.0])
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        if len(sys.argv) > 2 and sys.argv[2] not in name:
            continue
        key = (name[:90], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"], r.get("LDS_Block_Size", ""))
        rows[key][0] += 1
        rows[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print(f"{t / n:9.1f} us x {n:5d}  grid {k[1]}x{k[2]}x{k[3]} wg {k[4]} lds {k[5]}  {k[0]}")
