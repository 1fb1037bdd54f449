import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench as B
from drn_amd import dist as ddist, functional as DF, optim
from drn_amd.model import mainModel
from drn_amd.utils.synthetic import default_cfg, synthetic_batch
dev = torch.device("cuda:0")
cfg = default_cfg("C3D", 4096, 1)
batch = [b.to(dev) for b in synthetic_batch(32, 256, 4096, seed=1)]
for ext in (False, True):
    optim.EXT_SUMSQ = ext
    m = B.build(mainModel, cfg, dev, compute_dtype=torch.bfloat16)
    params = B.stage_params(m, 1)
    m.train()
    red = ddist.GradReducer(params, world_size=1, overlap=True, adjacent=m.grad_stack_groups(), bucket_bytes=1 << 30)
    opt = optim.FusedAdam(red, lr=1e-3, max_norm=0.5)
    for it in range(2):
        red.zero()
        _, ls = m(*batch)
        DF.backward(DF.loss_total(ls))
        red.finish()
        flat = red.buckets[0].flat
        true = float((flat.double() ** 2).sum())
        notes = getattr(red, "sumsq_notes", [])
        opt.norm()
        torch.cuda.synchronize()
        print("ext", ext, "step", it, "true", true, "kernel", float(opt.total_sumsq), "notes", [(len(nt[0]), nt[1].numel(), float(nt[1].double().sum())) for nt in notes])
        if ext:
            base = flat.data_ptr()
            for nt in notes:
                tot = 0.0
                for ptr, ne in nt[0]:
                    lo = (ptr - base) // 4
                    tot += float((flat[lo:lo + ne].double() ** 2).sum())
                print("   note ranges true sumsq", tot, "partials sum", float(nt[1].double().sum()), [((p - base) // 4, ne) for p, ne in nt[0]][:12])
        opt.update()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        opt.norm()
    e1.record()
    torch.cuda.synchronize()
    print("   norm() %.1f us" % (e0.elapsed_time(e1) * 1e3 / 20))
    red.remove()
