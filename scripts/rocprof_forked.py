#!/usr/bin/env python
"""Timeline of the LAST replay of the two-branch (forked) step in a rocprofv3 rocpd DB: every kernel with its start offset,
duration, and how many other kernels of the step were running at its start (0 = alone on the chip) -- which launches of the
query side really run beside the main branch, and which main-branch launches stretch when they do.
usage: python scripts/rocprof_forked.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end, grid_x, workgroup_x, grid_y, grid_z from kernels order by start"))
# a replay ends with the last kernel of the optimizer / repack phase: cut between an adam_tiled kernel and the next pos_feat / qe_embed
# a replay = from the first kernel of one norm pass (sumsq_partials burst) to the first of the next: the optimizer-first order starts
# with it; in the classic order (norm pass near the end) the window is the same steady-state cycle, rotated
marks = [i for i, r in enumerate(rows) if r[0].startswith("sumsq_partials_kernel")]
starts = [i for k, i in enumerate(marks) if k == 0 or marks[k - 1] != i - 1]
lo, hi = starts[-2], starts[-1]
step = rows[lo:hi]
t0 = min(r[1] for r in step)
QUERY = ("qe_", "lstm_", "skinny_group", "outer_wgrad", "colsum_partial", "colsum_segs")
busy = 0.0
events = sorted([(r[1], 1) for r in step] + [(r[2], -1) for r in step])
# union of busy intervals
cur, last, union = 0, None, 0.0
for t, d in events:
    if cur > 0:
        union += t - last
    cur += d
    last = t
side_total = main_total = 0.0
for i, (n, s, e, gx, wx, gy, gz) in enumerate(sorted(step, key=lambda r: r[1])):
    wgs = (gx // max(wx, 1)) * max(gy, 1) * max(gz, 1)
    others = sum(1 for r in step if r[1] <= s < r[2] and (r[1], r[2], r[0]) != (s, e, n))
    side = any(q in n for q in QUERY)
    if side:
        side_total += (e - s) / 1e3
    else:
        main_total += (e - s) / 1e3
    short = n.replace("void ", "")[:70]
    print("%4d %9.1f %8.1f  %s %d  %7d  %s" % (i, (s - t0) / 1e3, (e - s) / 1e3, "Q" if side else "M", others, wgs, short))
print("# %d kernels, span %.1f us, union of busy intervals %.1f us, main-branch kernel time %.1f us, query-side kernel time %.1f us"
      % (len(step), (max(r[2] for r in step) - t0) / 1e3, union / 1e3, main_total, side_total))
