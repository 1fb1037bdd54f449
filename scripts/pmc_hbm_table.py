#!/usr/bin/env python
"""Per-kernel HBM-side traffic table from three rocprofv3 runs of the same bench command (scripts/prof_round.sh):
  trace.db  --kernel-trace --stats       (clean durations)
  fetch.db  --pmc FETCH_SIZE             (its own pass, no trace domains beside --kernel-trace)
  write.db  --pmc WRITE_SIZE             (its own pass)
FETCH_SIZE / WRITE_SIZE are in KiB-ish units of 1024 B per the rocprofv3 derived-metric definition; FETCH_SIZE is DOUBLED (gfx950:
128-B requests tallied at 64 B for wide coalesced reads, MI355X_MICROARCH.md "HBM").  Infinity-Cache hits are counted: the
figure is fabric traffic behind the L2, an upper bound of HBM traffic.
usage: python scripts/pmc_hbm_table.py trace.db fetch.db write.db [pmc_traffic.json] > profiles/rNN_hbm_kernels.txt
With a 4th argument the per-launch bytes of the two prop_fc GEMMs (gemm_nt_w4_kernel's 512- and 256-workgroup launches) are
written there in the format bench.py reads for roofline.traffic."""
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


def by_grid(dbpath, counter, name_part):
    """{workgroups per launch: mean counter value} over the dispatches whose kernel name contains name_part."""
    db = sqlite3.connect(dbpath)
    cur = db.cursor()
    cur.execute("select * from counters_collection limit 1")
    cols = [d[0] for d in cur.description]
    ki, ci, vi = cols.index("kernel_name"), cols.index("counter_name"), cols.index("value")
    gcol = [c for c in cols if c in ("grid_size", "grid_size_x", "grid_x")]
    wcol = [c for c in cols if c in ("workgroup_size", "workgroup_size_x", "workgroup_x")]
    if not gcol or not wcol:
        sys.stderr.write("counters_collection has no grid / workgroup size columns: %s\n" % cols)
        return {}
    gi, wi = cols.index(gcol[0]), cols.index(wcol[0])
    agg = collections.defaultdict(list)
    for r in cur.execute("select * from counters_collection"):
        if r[ci] == counter and name_part in r[ki]:
            agg[int(r[gi]) // max(int(r[wi]), 1)].append(float(r[vi]))
    out = {}
    for k, v in agg.items():
        # two different launches can share a grid size (round 3: conv0's data gradient also runs 512 tiles of 256x256); the
        # prop_fc products are the larger ones on both counters: keep the upper cluster when the values are clearly bimodal
        lo, hi = min(v), max(v)
        if hi > 1.3 * lo:
            mid = 0.5 * (lo + hi)
            v = [x for x in v if x > mid]
        out[k] = sum(v) / len(v)
    return out


def write_traffic_json(path, fetch, write):
    import json
    big = "gemm_nt_w4_kernel"               # the two prop_fc products run the 4-wave kernel (round 4; before: "2, 4, 8, 4, false", the 256x256-tile NT kernel)
    F, W = by_grid(fetch, "FETCH_SIZE", big), by_grid(write, "WRITE_SIZE", big)
    out = {"_comment": "Fabric traffic per launch of the two prop_fc GEMMs INSIDE the replayed step, from this round's rocprofv3 --pmc passes "
                       "(FETCH_SIZE and WRITE_SIZE in separate passes, only --kernel-trace beside them; FETCH_SIZE doubled per the gfx950 "
                       "correction of MI355X_MICROARCH.md), scripts/prof_round.sh -> scripts/pmc_hbm_table.py.  bench.py copies the entry of "
                       "its dominant kernel into roofline.traffic."}
    for wgs, tag, alg in ((512, "gemm_nt[bf16] g=1 M=8192 N=4096 K=4096 mode=0", 2 * (8192 * 4096 + 4096 * 4096 + 2 * 8192 * 4096)),   # x, W; gated output + pre-gate copy
                          (256, "gemm_nt[bf16] g=1 M=4096 N=4096 K=8192 mode=0", 2 * 2 * 4096 * 8192 + 4 * 4096 * 4096)):
        if wgs in F and wgs in W:
            out[tag] = {"read_bytes": int(2 * 1024 * F[wgs]), "write_bytes": int(1024 * W[wgs]), "algorithmic_bytes": alg,
                        "served_by": "Infinity Cache + HBM behind the eight private L2s (32 tiles per XCD at a time touch >= 24 MB of operand "
                                     "panels, i.e. >= 384 MB per launch, DESIGN.md section 5); compulsory HBM bytes = algorithmic"}
    with open(path, "w") as f:
        json.dump(out, f, indent=1)


def main():
    trace, fetch, write = sys.argv[1:4]
    if len(sys.argv) > 4:
        write_traffic_json(sys.argv[4], fetch, write)
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
