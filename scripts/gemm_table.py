#!/usr/bin/env python
"""Per-GEMM table of ONE replayed step: joins the MFMA launches of a step in launch order (bench.py --dump-gemms: tag with the
shape, FLOPs) with the durations of the same launches inside the replayed hipGraph (rocprofv3 --kernel-trace rocpd DB).
usage: python scripts/gemm_table.py <results.db> <gemms.json> [peak TFLOP/s = 2500] [out.json: {tag: {us, TFLOP/s, frac}}, read by
bench.py for `roofline.in_graph`]"""
import json
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
gemms = json.load(open(sys.argv[2]))
peak = float(sys.argv[3]) if len(sys.argv) > 3 else 2500.0
rows = list(db.execute("select name, start, end, grid_x, workgroup_x, grid_y, grid_z from kernels order by start"))
marks = [i for i, r in enumerate(rows) if r[0].startswith("adam_bucket_kernel")]
bursts = [i for k, i in enumerate(marks) if k + 1 == len(marks) or marks[k + 1] != i + 1]
step = rows[bursts[-2] + 1:bursts[-1] + 1]
is_gemm = lambda n: "conv_gemm_nt_kernel" in n or "conv_wgrad" in n or "gemm_nt_w4_kernel" in n or "gemm_nt_w4c_kernel" in n or "gemm_nt_w4h_kernel" in n
# a weight gradient with a separate reduce pass is ONE tagged launch followed by its wgrad_reduce kernel(s): fold them in
launches = []
reduce_all = 0.0
for n, s, e, gx, wx, gy, gz in step:
    if is_gemm(n):
        launches.append([n, (e - s) / 1e3, (gx // max(wx, 1)) * max(gy, 1) * max(gz, 1), 0.0])
    elif n.startswith("wgrad_reduce_all"):           # the step's deferred reduce passes in one launch: its own line below
        reduce_all = (e - s) / 1e3
    elif n.startswith("wgrad_reduce") and launches:
        launches[-1][3] += (e - s) / 1e3
if len(launches) != len(gemms):
    sys.exit("launch count mismatch: %d MFMA kernels in the trace step, %d tagged launches" % (len(launches), len(gemms)))
print("# MFMA launches of one replayed step (durations inside the hipGraph), peak %.0f TFLOP/s" % peak)
print("%-3s %-58s %7s %8s %8s %7s %6s" % ("#", "launch", "wgs", "us", "+reduce", "TF/s", "%peak"))
tot_us = tot_fl = 0.0
for i, ((tag, fl), (n, us, wgs, red)) in enumerate(zip(gemms, launches)):
    tf = fl / (us * 1e-6) / 1e12
    tot_us += us + red
    tot_fl += fl
    print("%-3d %-58s %7d %8.1f %8.1f %7.0f %5.1f%%" % (i, tag[:58], wgs, us, red, tf, 100.0 * tf / peak))
if len(sys.argv) > 4:
    out = {}
    for (tag, fl), (n, us, wgs, red) in zip(gemms, launches):
        out[tag] = {"us": round(us, 1), "reduce_us": round(red, 1), "TFLOP/s": round(fl / (us * 1e-6) / 1e12, 1), "frac": round(fl / (us * 1e-6) / 1e12 / peak, 4)}
    json.dump(out, open(sys.argv[4], "w"), indent=0)
if reduce_all:
    print("#   + wgrad_reduce_all_kernel (every weight gradient's reduce pass, one launch) %.1f us" % reduce_all)
    tot_us += reduce_all
print("# total %.1f us (reduce passes included), %.1f GFLOP, %.0f TFLOP/s = %.1f %% of peak"
      % (tot_us, tot_fl / 1e9, tot_fl / (tot_us * 1e-6) / 1e12, 100.0 * tot_fl / (tot_us * 1e-6) / 1e12 / peak))
