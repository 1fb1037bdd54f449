#!/usr/bin/env python
"""Per-kernel HBM-side traffic table from three rocprofv3 runs of the same bench command (scripts/prof_round.sh):
  trace.db  --kernel-trace --stats       (clean durations)
  fetch.db  --pmc FETCH_SIZE             (its own pass, no trace domains beside --kernel-trace)
  write.db  --pmc WRITE_SIZE             (its own pass)
FETCH_SIZE / WRITE_SIZE are in KiB-ish units of 1024 B per the rocprofv3 derived-metric definition; FETCH_SIZE is DOUBLED (gfx950:
128-B requests tallied at 64 B for wide coalesced reads, MI355X_MICROARCH.md "HBM").  Infinity-Cache hits are counted: the
figure is fabric traffic behind the L2, an upper bound of HBM traffic.
usage: python scripts/pmc_hbm_table.py trace.db fetch.db write.db [steps_in_trace] > profiles/rNN_hbm_kernels.txt"""
import collections
import sqlite3
import sys


def short(n):
    n = n.replace("void ", "").replace("at::native::", "at::")
    for cut in ("(", "<"):
        if cut in n and not n.startswith("_Z"):
            n = n[:n.index(cut)] if cut == "(" else n
    return n[:64]


def key(n):
    """Kernel family name without arguments (template arguments kept for the GEMMs)."""
    if n.startswith("_Z"):
        import re
        m = re.match(r"_Z\d+([A-Za-z0-9_]+?)I", n)
        return (m.group(1) if m else n)[:60] + ("<bf16>" if "DF16b" in n else "")
    n = n.replace("void ", "")
    return n.split("(")[0][:60]


def pmc(dbpath, counter):
    db = sqlite3.connect(dbpath)
    cur = db.cursor()
    cur.execute("select * from counters_collection limit 1")
    cols = [d[0] for d in cur.description]
    ki, ci, vi = cols.index("kernel_name"), cols.index("counter_name"), cols.index("value")
    agg = collections.defaultdict(list)
    for r in cur.execute("select * from counters_collection"):
        if r[ci] == counter:
            agg[key(r[ki])].append(float(r[vi]))
    return agg


def main():
    trace, fetch, write = sys.argv[1:4]
    db = sqlite3.connect(trace)
    dur = collections.defaultdict(list)
    for name, s, e in db.execute("select name, start, end from kernels"):
        dur[key(name)].append((e - s) / 1e3)
    F, W = pmc(fetch, "FETCH_SIZE"), pmc(write, "WRITE_SIZE")
    unit = 1024.0          # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB
    print("# fabric-side bytes per launch (rocprofv3 --pmc, separate passes) and the rate they imply at the clean-trace duration")
    print("# FETCH_SIZE x2 (gfx950 correction), WRITE_SIZE as reported; peak HBM 8000 GB/s; Infinity-Cache hits are included in the bytes")
    print("%-58s %7s %9s %11s %11s %9s %7s" % ("kernel", "calls", "avg_us", "read_MB", "write_MB", "GB/s", "of 8TB/s"))
    rows = []
    for k, ds in dur.items():
        if k not in F and k not in W:
            continue
        f = 2.0 * unit * (sum(F[k]) / len(F[k])) if F.get(k) else 0.0
        w = unit * (sum(W[k]) / len(W[k])) if W.get(k) else 0.0
        avg = sum(ds) / len(ds)
        rows.append((sum(ds), k, len(ds), avg, f, w, (f + w) / (avg * 1e-6) / 1e9))
    for tot, k, n, avg, f, w, gbs in sorted(rows, reverse=True)[:45]:
        print("%-58s %7d %9.1f %11.2f %11.2f %9.0f %6.1f%%" % (k, n, avg, f / 1e6, w / 1e6, gbs, gbs / 80.0))


main()
