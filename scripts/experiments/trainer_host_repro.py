import faulthandler, os, sys, time
faulthandler.enable()
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench as B_
from drn_amd import trainer as TR
from drn_amd.model import mainModel
from drn_amd.utils.synthetic import default_cfg, synthetic_batch
dev = torch.device("cuda:0")
B, T, D, stage = 32, int(os.environ.get("T", "256")), 4096, 1
cfg = default_cfg("C3D", D, stage)
batches = [B_.collate_like([t.to(dev) if torch.is_tensor(t) else t for t in synthetic_batch(B, T, D, seed=100 + i)], ["v%d" % i] * B) for i in range(8)]
m = B_.build(mainModel, cfg, dev, compute_dtype=torch.bfloat16)
tr = TR.Trainer(m, stage, lr=1e-3, clip_gradient=0.5, graph=True, forked=os.environ.get("FORKED", "auto") if os.environ.get("FORKED", "auto") == "auto" else os.environ["FORKED"] == "1")
if os.environ.get("DEVICE_FIRST", "0") == "1":
    for _ in range(3):
        tr.train_epoch(batches)
    print("device-resident epochs done", flush=True)
fdt = torch.float32 if os.environ.get("FDT") == "f32" else torch.bfloat16
hb = batches if os.environ.get("KEEP_DEV") == "1" else [tuple((t.to(fdt) if i == 2 else t).cpu().pin_memory() if torch.is_tensor(t) else t for i, t in enumerate(b)) for b in batches]
for e in range(4):
    tr.train_epoch(hb)
    torch.cuda.synchronize()
    print("host epoch", e, "done; slots:", {k: (s.graph is not None, s.fork_done, s.fork_log) for k, s in tr._slots.items()}, flush=True)
t0 = time.perf_counter()
tr.train_epoch(hb * 8)
torch.cuda.synchronize()
print("host bf16 inputs: %.3f ms/step" % ((time.perf_counter() - t0) / 64 * 1e3), flush=True)
