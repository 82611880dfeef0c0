#!/usr/bin/env python3
"""Per (kernel, grid) launch counts and average durations of a rocprofv3 kernel trace (rocpd sqlite).  prof_by_grid.py db [substr ...]"""
import re, sqlite3, sys
def short(n):
    n = n.replace('(anonymous namespace)::', ''); n = re.sub(r'isi::', '', n); n = re.sub(r'\(.*', '', n); return n[:60]
db = sqlite3.connect(sys.argv[1]); want = sys.argv[2:]
g = {}
for name, st, en, gx, gy, gz, wx in db.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start"):
    k = short(name)
    if want and not any(w in k for w in want): continue
    a = g.setdefault((k, gx // max(wx, 1), gy, gz), [0, 0.0]); a[0] += 1; a[1] += (en - st) / 1e3
for k, v in sorted(g.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[0]:60s} grid=({k[1]},{k[2]},{k[3]}) calls {v[0]:7d} avg {v[1] / v[0]:8.2f} us total {v[1] / 1e3:9.2f} ms")
