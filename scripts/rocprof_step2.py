#!/usr/bin/env python
"""Kernel timeline of the LAST full bench step in a rocprofv3 rocpd DB when the step runs on several streams: start offset, end
offset, duration, queue/stream tag, workgroups, name -- ordered by start.  A '*' marks kernels that overlap another stream's.
usage: python scripts/rocprof_step2.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
rows = list(db.execute("select name, start, end, grid_x, workgroup_x, grid_y, grid_z, %s from kernels order by start" % (qcol or "0")))
marks = [i for i, r in enumerate(rows) if r[0].startswith("adam_bucket_kernel")]
lo, hi = marks[-2] + 1, marks[-1] + 1
step = rows[lo:hi]
t0 = step[0][1]
qs = sorted(set(r[7] for r in step))
busy = {}
for i, (n, s, e, gx, wx, gy, gz, q) in enumerate(step):
    wgs = (gx // max(wx, 1)) * max(gy, 1) * max(gz, 1)
    ov = any(o[7] != q and o[1] < e and o[2] > s for o in step)
    busy[q] = busy.get(q, 0.0) + (e - s) / 1e3
    short = n.replace("void at::native::", "at::")[:70]
    print("%4d %9.1f %9.1f %8.1f  q%-2d %s %7d  %s" % (i, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, qs.index(q), "*" if ov else " ", wgs, short))
print("# %d kernels, span %.1f us, busy per queue %s (column %s)" % (len(step), (max(r[2] for r in step) - t0) / 1e3,
                                                                   {qs.index(k): round(v, 1) for k, v in busy.items()}, qcol))
