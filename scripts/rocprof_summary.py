#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite output) as a text table.
usage: python scripts/rocprof_summary.py <results.db> [steps] > profiles/<name>.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
unit = 1e-3  # top_kernels durations are in microseconds? check scale below
tot = sum(r[2] for r in rows)
print("# rocprofv3 --kernel-trace --stats summary (%s); durations in us%s" % (sys.argv[1].split('/')[-1], ", %d bench steps incl. warm-up" % steps if steps else ""))
print("# total kernel time %.1f us over %d kernels%s" % (tot, len(rows), ("  = %.1f us/step" % (tot / steps)) if steps else ""))
print("%-78s %8s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "%"))
for name, calls, total, avg, pct in rows[:60]:
    short = name
    if short.startswith("Cijk_"):
        short = short[:40] + "...(hipBLASLt)"
    short = short.replace("void at::native::", "at::").replace("(anonymous namespace)::", "")
    if len(short) > 76:
        short = short[:73] + "..."
    print("%-78s %8d %12.1f %10.2f %6.2f" % (short, calls, total, avg, pct))
