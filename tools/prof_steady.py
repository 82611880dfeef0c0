#!/usr/bin/env python3
"""Per-kernel time of ONE steady-state training step from a rocprofv3 kernel trace (rocpd sqlite): steps are delimited by the
last dispatch of the optimizer kernel (name contains `marker`); the first `skip` steps (warm-up: packs, first-use set-up)
are left out and the rest is averaged.   prof_steady.py results.db [marker] [skip]"""
import re, sqlite3, sys


def short(n):
    n = n.replace('(anonymous namespace)::', '')
    n = re.sub(r'isi::', '', n)
    n = re.sub(r'\(.*', '', n)
    return n[:72]


def main(path, marker="multi_tensor_apply", skip=2, grids=()):
    db = sqlite3.connect(path)
    rows = list(db.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start"))
    ends, prev_marker = [], False
    for i, r in enumerate(rows):          # a step ends with its LAST consecutive-ish optimizer dispatch
        is_m = marker in r[0]
        if prev_marker and not is_m:
            ends.append(i)                # index of the first dispatch after the optimizer group
        prev_marker = is_m
    if prev_marker:
        ends.append(len(rows))
    # optimizer groups closer than 20 dispatches belong to one step (several launches of the fused optimizer)
    merged = []
    for e in ends:
        if merged and e - merged[-1] < 20:
            merged[-1] = e
        else:
            merged.append(e)
    ends = merged
    if len(ends) <= skip:
        print("not enough steps", len(ends)); return
    lo, hi, nsteps = ends[skip - 1] if skip > 0 else 0, ends[-1], len(ends) - skip
    agg = {}
    for r in rows[lo:hi]:
        a = agg.setdefault(short(r[0]), [0, 0.0])
        a[0] += 1; a[1] += (r[2] - r[1]) / 1e3
    tot = sum(v[1] for v in agg.values())
    wall = (rows[hi - 1][2] - rows[lo][1]) / 1e3
    print(f"# {nsteps} steady-state steps: {wall / nsteps:.1f} us wall per step, {tot / nsteps:.1f} us of kernels per step")
    print(f"{'kernel':72s} {'calls/step':>10s} {'us/step':>10s} {'avg_us':>9s} {'pct':>6s}")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:72s} {v[0] / nsteps:10.1f} {v[1] / nsteps:10.1f} {v[1] / v[0]:9.1f} {100 * v[1] / tot:6.2f}")
    if grids:
        print("\n# launches by grid (workgroups x, y, z): calls/step, avg us")
        g = {}
        for r in rows[lo:hi]:
            k = short(r[0])
            if any(s_ in k for s_ in grids):
                a = g.setdefault((k, r[3] // max(r[6], 1), r[4], r[5]), [0, 0.0])
                a[0] += 1; a[1] += (r[2] - r[1]) / 1e3
        for k, v in sorted(g.items(), key=lambda kv: (kv[0][0], -kv[1][1])):
            print(f"{k[0]:72s} grid=({k[1]},{k[2]},{k[3]}) {v[0] / nsteps:6.1f} {v[1] / v[0]:9.1f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "multi_tensor_apply", int(sys.argv[3]) if len(sys.argv) > 3 else 2, tuple(sys.argv[4:]))
