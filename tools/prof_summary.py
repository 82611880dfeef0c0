#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite (kernel-trace) into per-kernel stats and the
per-dispatch timeline of the last forward pass (text)."""
import re
import sqlite3
import sys


def short(n):
    n = n.replace('(anonymous namespace)::', '')
    n = re.sub(r'isi::', '', n)
    n = re.sub(r'\(.*', '', n)
    return n[:64]


def main(path, last=40):
    db = sqlite3.connect(path)
    rows = list(db.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start"))
    agg = {}
    for r in rows:
        a = agg.setdefault(short(r[0]), [0, 0.0])
        a[0] += 1
        a[1] += (r[2] - r[1]) / 1e3
    tot = sum(v[1] for v in agg.values())
    print(f"{'kernel':64s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:64s} {v[0]:6d} {v[1]:12.1f} {v[1] / v[0]:10.1f} {100 * v[1] / tot:6.2f}")
    if last <= 0:
        return
    print("\nlast dispatches:")
    for r in rows[-last:]:
        wg = max(r[6], 1)
        print(f"  {short(r[0]):64s} {(r[2] - r[1]) / 1e3:9.1f} us grid=({r[3] // wg},{r[4]},{r[5]})")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
