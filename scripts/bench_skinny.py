"""Micro-benchmark of drn_skinny_group (through ops.skinny_linear, one problem per launch) on the query-side shapes (M = 32 clips)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from drn_amd import ops
shapes = [("qInput", 512, 2048), ("qInput012", 3072, 512), ("dbase", 512, 3072), ("dqvec", 2048, 512), ("gate0", 4096, 1024),
          ("gate1", 256, 1024), ("gate2", 512, 1024), ("dgate0", 1024, 4096), ("dgate1", 1024, 256), ("dgate2", 1024, 512)]
tot = 0.0
for name, N, K in shapes:
    x = torch.randn(32, K, device="cuda"); W = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
    for _ in range(5): ops.skinny_linear(x, W, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.skinny_linear(x, W, b)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 20
    tot += us
    print("%-10s N=%5d K=%5d  %6.1f us" % (name, N, K, us))
print("total %.1f us" % tot)
