import sys, time, torch
sys.path.insert(0, '.')
import bench as B
from drn_amd import trainer as TR
from drn_amd.model import mainModel
from drn_amd.utils.synthetic import default_cfg, synthetic_batch
dev = torch.device("cuda:0")
cfg = default_cfg("C3D", 4096, 1)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 256
batches = [B.collate_like([t.to(dev) if torch.is_tensor(t) else t for t in synthetic_batch(32, T, 4096, seed=100 + i)], ["v%d" % i] * 32) for i in range(8)]
for forked in (False, True, False, True):
    m = B.build(mainModel, cfg, dev, compute_dtype=torch.bfloat16)
    tr = TR.Trainer(m, 1, lr=1e-3, clip_gradient=0.5, graph=True, forked=forked)
    for _ in range(3):
        tr.train_epoch(batches)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(6):
        tr.train_epoch(batches)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 48
    print("forked=%s: %.3f ms/step, slots %s" % (forked, dt * 1e3, [(k, s.graph is not None, s.forked is not None) for k, s in tr._slots.items()]), flush=True)
    tr.reducer.remove()
    del m, tr
