#!/usr/bin/env python
"""Print the kernel sequence of the LAST full bench step in a rocprofv3 rocpd DB (between the last two
sumsq_partials bursts): index, start offset, duration, gap to the previous kernel, workgroups, name.
usage: python scripts/rocprof_step.py <results.db> [name-filter]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else None
rows = list(db.execute("select name, start, end, grid_x, workgroup_x, grid_y, grid_z from kernels order by start"))
marks = [i for i, r in enumerate(rows) if r[0].startswith("adam_bucket_kernel")]
# last step = after the 2nd-to-last adam burst's final kernel up to the last burst's final kernel
bursts = [i for k, i in enumerate(marks) if k + 1 == len(marks) or marks[k + 1] != i + 1]
lo, hi = bursts[-2] + 1, bursts[-1] + 1
step = rows[lo:hi]
t0 = step[0][1]
prev_end = t0
tot = gap_tot = 0.0
for i, (n, s, e, gx, wx, gy, gz) in enumerate(step):
    wgs = (gx // max(wx, 1)) * max(gy, 1) * max(gz, 1)
    gap = (s - prev_end) / 1e3
    prev_end = max(prev_end, e)
    tot += (e - s) / 1e3
    gap_tot += max(gap, 0)
    short = n.replace("void at::native::", "at::").replace("(anonymous namespace)::", "")[:90]
    if flt is None or flt in n:
        print("%4d %9.1f %8.1f %7.1f %7d  %s" % (i, (s - t0) / 1e3, (e - s) / 1e3, gap, wgs, short))
print("# %d kernels, busy %.1f us, gaps %.1f us, span %.1f us" % (len(step), tot, gap_tot, (step[-1][2] - t0) / 1e3))
