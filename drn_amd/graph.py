"""hipGraph capture of a whole training step (forward + backward + optimizer), MI355X-style replacement for a
tracing compiler: the step's ~300 kernel launches are recorded once on a capture stream and replayed with one
host call, so the Python/ctypes launch path disappears from the steady state.

Requirements on `step_fn` (all met by the DRN step in bench.py): static shapes, inputs read from fixed device
buffers, no host synchronisation (`.item()`, `nonzero`, host-dependent control flow), gradients accumulated into
persistent buffers (drn_amd.dist.GradReducer), optimizer state on the device (drn_amd.optim.FusedAdam).
"""
import torch


class GraphedStep(object):
    def __init__(self, step_fn, warmup=3):
        self.step_fn = step_fn
        self.graph = None
        self.out = None
        # warm up on a side stream (allocator pools, MIOpen/hipBLASLt handles, workspaces, lazy HIP module loads)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self.out = step_fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()

    def capture(self):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.out = self.step_fn()
        self.graph = g
        return self

    def __call__(self):
        if self.graph is None:
            return self.step_fn()
        self.graph.replay()
        return self.out
