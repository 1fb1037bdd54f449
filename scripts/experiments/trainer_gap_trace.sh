# GPU timeline of the Trainer loop between two replays (device-resident batches, T = 32): what sits between the graphs
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_trainer
mkdir -p $OUT
cat > /tmp/tl.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
import bench as B
from drn_amd import trainer as TR
from drn_amd.model import mainModel
from drn_amd.utils.synthetic import default_cfg, synthetic_batch
dev = torch.device("cuda:0")
T = int(sys.argv[1])
cfg = default_cfg("C3D", 4096, 1)
bs = [B.collate_like([t.to(dev) if torch.is_tensor(t) else t for t in synthetic_batch(32, T, 4096, seed=100 + i)], ["v%d" % i] * 32) for i in range(8)]
m = B.build(mainModel, cfg, dev, compute_dtype=torch.bfloat16)
tr = TR.Trainer(m, 1, lr=1e-3, clip_gradient=0.5, graph=True)
for _ in range(5):
    tr.train_epoch(bs)
torch.cuda.synchronize()
PY
for T in ${TLIST:-32}; do
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/trace -o t -- python /tmp/tl.py $T > /dev/null 2>&1
python - <<PY
import sqlite3
db = sqlite3.connect("$OUT/trace/t_results.db")
rows = list(db.execute("select name, start, end from kernels order by start"))
# steps are delimited by adam_bucket_kernel (last kernel of a replay)
idx = [i for i, r in enumerate(rows) if r[0].startswith("adam_bucket_kernel")]
a, b = idx[-3], idx[-2]
seg = rows[a:b + 1]
t0 = seg[0][1]
print("T=$T: one trainer step, from the end of a replay to the end of the next: %.1f us" % ((seg[-1][2] - seg[0][2]) / 1e3))
prev = seg[0][2]
for n, s, e in seg[1:int("${NSHOW:-14}")]:
    print("  +%7.1f  gap %6.1f  dur %6.1f  %s" % ((s - seg[0][2]) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, n[:60]))
    prev = e
print("  ...")
PY
done
rm -rf $OUT/trace
