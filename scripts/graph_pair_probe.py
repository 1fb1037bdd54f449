"""Two streams: a linear hipGraph on one, and on the other either a second graph or the same kernels launched eagerly -- when
does the second stream's first kernel start?  Run under rocprofv3 --kernel-trace and read the DB (scripts/_r2k.sh).
argv: n1 n2 mode(graph|eager_after|eager_before)"""
import sys
import torch
n1, n2, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
dev = torch.device("cuda", 0)
a = torch.zeros(1 << (int(sys.argv[4]) if len(sys.argv) > 4 else 20), device=dev)
b = torch.zeros(1 << (int(sys.argv[4]) if len(sys.argv) > 4 else 20), device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def body(t, n):
    for _ in range(n):
        t.add_(1.0)
with torch.cuda.stream(s1):
    body(a, 3)
with torch.cuda.stream(s2):
    body(b, 3)
torch.cuda.synchronize()
g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
with torch.cuda.graph(g1, stream=s1):
    body(a, n1)
with torch.cuda.graph(g2, stream=s2):
    body(b, n2)
c = torch.zeros(1 << 24, device=dev)
d = torch.zeros(1 << 26, device=dev)
for it in range(6):
    torch.cuda.synchronize()
    c.mul_(1.0)            # marker kernel (different size) on the default stream
    torch.cuda.synchronize()
    # a long kernel first on both streams so that everything below is enqueued before any of it can start
    ev = torch.cuda.Event()
    d.add_(1.0); d.add_(1.0); d.add_(1.0); d.add_(1.0)
    ev.record()
    s1.wait_event(ev); s2.wait_event(ev)
    if mode == "eager_before":
        with torch.cuda.stream(s2):
            body(b, n2)
    with torch.cuda.stream(s1):
        g1.replay()
    if mode == "graph":
        with torch.cuda.stream(s2):
            g2.replay()
    elif mode == "eager_after":
        with torch.cuda.stream(s2):
            body(b, n2)
torch.cuda.synchronize()
