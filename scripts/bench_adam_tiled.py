"""Micro-benchmark of drn_adam_tiled by tensor shape / copy set: us per launch and the HBM rate its algorithmic bytes imply
(28 B per parameter + the copies).  usage (GPU box): python scripts/bench_adam_tiled.py"""
import ctypes
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drn_amd._lib import AdamTiledItem, check, lib

dev = torch.device("cuda:0")
L = lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None


def case(name, R, C, k, c1, c2, reps=30, ncopies_of_tensor=8):
    """c1 / c2: None, 'bf16' or 'f32' copy of kind 1 / kind 2; the tensor is replicated `ncopies_of_tensor` times (separate items)
    so that the launch is large enough to stream."""
    n = R * C * k
    N = ncopies_of_tensor
    g = torch.randn(N * n, device=dev) * 1e-3
    m = torch.zeros_like(g); v = torch.zeros_like(g)
    ps = [torch.randn(R, C * k, device=dev) for _ in range(N)]
    dt = {"bf16": torch.bfloat16, "f32": torch.float32}
    items, keep, blk_item, blk_tile = [], [], [], []
    for i, p in enumerate(ps):
        it = AdamTiledItem(p=p.data_ptr(), off=i * n, m1=None, m2=None, ld1=0, ld2=0, R=R, C=C, k=k, code1=0, code2=0, tiles_c=(C + 63) // 64)
        if c1:
            b = torch.empty(R, k, C, dtype=dt[c1], device=dev); keep.append(b)
            it.m1, it.ld1, it.code1 = b.data_ptr(), C, int(c1 == "bf16")
        if c2:
            b = torch.empty(C, k, R, dtype=dt[c2], device=dev); keep.append(b)
            it.m2, it.ld2, it.code2 = b.data_ptr(), R, int(c2 == "bf16")
        nt = ((R + 63) // 64) * it.tiles_c
        blk_item += [i] * nt; blk_tile += list(range(nt)); items.append(it)
    arr = (AdamTiledItem * len(items))(*items)
    raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    bi = torch.tensor(blk_item, dtype=torch.int32, device=dev); bt = torch.tensor(blk_tile, dtype=torch.int32, device=dev)
    tot = torch.ones(1, device=dev); step = torch.ones(1, dtype=torch.int32, device=dev)
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    def run():
        check(L.drn_adam_tiled(P(g), P(m), P(v), P(raw), P(bi), P(bt), len(blk_item), P(tot), P(step), ctypes.c_float(1e-3),
                               ctypes.c_float(0.9), ctypes.c_float(0.999), ctypes.c_float(1e-8), ctypes.c_float(0.5), ctypes.c_float(1.0), s), "adam_tiled")
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    bpp = 28 + sum({"bf16": 2, "f32": 4}[c] for c in (c1, c2) if c)
    print("%-44s %8.1fM params %4d wgs %8.1f us  %6.0f GB/s (%d B/param)" % (name, N * n / 1e6, len(blk_item), us, N * n * bpp / us / 1e3, bpp))


case("conv0 (256,4352,3) bf16+bf16", 256, 4352, 3, "bf16", "bf16", ncopies_of_tensor=4)
case("conv0 no copies", 256, 4352, 3, None, None, ncopies_of_tensor=4)
case("tower (512,512,3) bf16+bf16", 512, 512, 3, "bf16", "bf16", ncopies_of_tensor=16)
case("1x1 (512,1024,1) bf16+bf16", 512, 1024, 1, "bf16", "bf16", ncopies_of_tensor=16)
case("linear (4096,1024) f32 kind2", 4096, 1024, 1, None, "f32", ncopies_of_tensor=3)
case("linear (4096,1024) no copies", 4096, 1024, 1, None, None, ncopies_of_tensor=3)
case("linear (1024,512) f32 kind1+kind2", 1024, 512, 1, "f32", "f32", ncopies_of_tensor=16)
case("W_hh (2048,512) f32 kind2", 2048, 512, 1, None, "f32", ncopies_of_tensor=8)
case("W_ih (2048,300) f32 kind1+2", 2048, 300, 1, "f32", "f32", ncopies_of_tensor=8)
