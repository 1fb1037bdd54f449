# Trainer loop at T = 256 fed from pinned bf16 host batches, linear vs two-branch graph: where does the two-branch step lose its time?
cd /tmp && export TMPDIR=/tmp
for F in 0 1; do
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_trh$F
mkdir -p $OUT
DRN_TRAINER_FORKED=$F timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/trace -o t -- python $GRAFT_REPO_ROOT/scripts/experiments/trainer_timeline_probe.py host > /dev/null 2>&1
python - <<PY
import sqlite3
db = sqlite3.connect("$OUT/trace/t_results.db")
rows = list(db.execute("select name, start, end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if r[0].startswith("adam_bucket_kernel")]
a, b = idx[-4], idx[-3]
seg = rows[a:b + 1]
print("forked=$F: step (adam_bucket end to adam_bucket end) %.1f us, %d kernels" % ((seg[-1][2] - seg[0][2]) / 1e3, len(seg) - 1))
# union busy time and the five largest idle gaps
ev = sorted([(s, 1) for n, s, e in seg[1:]] + [(e, -1) for n, s, e in seg[1:]])
cur, last, gaps = 0, seg[0][2], []
for t, d in ev:
    if cur == 0 and t > last:
        gaps.append((t - last, last))
    cur += d
    last = t if cur == 0 else last
gaps.sort(reverse=True)
for g, at in gaps[:6]:
    nxt = [n for n, s, e in seg[1:] if s >= at + g - 1][:1]
    print("   idle %7.1f us at +%8.1f us, then %s" % (g / 1e3, (at - seg[0][2]) / 1e3, (nxt[0][:50] if nxt else "?")))
try:
    cps = list(db.execute("select start, end from memory_copies order by start"))
    w0, w1 = seg[0][2], seg[-1][2]
    inwin = [(s, e) for s, e in cps if e > w0 and s < w1]
    print("   %d memory copies overlap the step, total %.1f us" % (len(inwin), sum(e - s for s, e in inwin) / 1e3))
except Exception as ex:
    print("   (no memory copy table: %s)" % ex)
PY
rm -rf $OUT
done
