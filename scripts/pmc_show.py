import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
try:
    rows = list(cur.execute("select * from counters_collection limit 1"))
    cols = [d[0] for d in cur.description]
except Exception as e:
    print("err", e); sys.exit()
ki = cols.index("kernel_name") if "kernel_name" in cols else None
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in cur.execute("select * from counters_collection"):
    d = dict(zip(cols, r))
    if "conv_" not in str(d.get("kernel_name", "")): continue
    agg[d["kernel_name"][:30]][d["counter_name"]].append(d["value"])
for k, v in agg.items():
    print(k)
    for c, vals in sorted(v.items()):
        print("   %-32s n=%d  last=%.4g" % (c, len(vals), vals[-1]))
