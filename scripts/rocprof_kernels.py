#!/usr/bin/env python
"""Per-kernel-name avg durations grouped by (name, grid) from a rocprofv3 rocpd DB."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else "%"
rows = db.execute("select name, start, end, grid_x, workgroup_x, grid_y, grid_z from kernels where name like ? order by start", (pat,))
seq = []
for n, s, e, gx, wx, gy, gz in rows:
    seq.append((n[:40], (gx // max(wx, 1)) * max(gy, 1) * max(gz, 1), (e - s) / 1e3))
# group consecutive identical (name, wgs)
out = []
for n, w, d in seq:
    if out and out[-1][0] == n and out[-1][1] == w:
        out[-1][2].append(d)
    else:
        out.append([n, w, [d]])
for n, w, ds in out:
    ds2 = sorted(ds)[len(ds) // 4: max(len(ds) * 3 // 4, len(ds) // 4 + 1)]
    print("%-42s wgs %6d  n=%3d  median-ish %8.1f us" % (n, w, len(ds), sum(ds2) / len(ds2)))
