"""hipGraph capture of a whole training step (forward + backward + optimizer), MI355X-style replacement for a
tracing compiler: the step's ~300 kernel launches are recorded once on a capture stream and replayed with one
host call, so the Python/ctypes launch path disappears from the steady state.

Requirements on `step_fn` (all met by the DRN step in bench.py): static shapes, inputs read from fixed device
buffers, no host synchronisation (`.item()`, `nonzero`, host-dependent control flow), gradients accumulated into
persistent buffers (drn_amd.dist.GradReducer), optimizer state on the device (drn_amd.optim.FusedAdam).

Every warm-up step and the capture itself run on ONE dedicated side stream: autograd's AccumulateGrad nodes (kept
alive by the reducer's hooks) remember the stream they were created on, and a node created on the default stream
would run -- uncaptured and unordered -- outside the graph.  So do not run `step_fn` eagerly on another stream first.
"""
import torch


class GraphedStep(object):
    def __init__(self, step_fn, warmup=3):
        self.step_fn = step_fn
        self.graph = None
        self.out = None
        self.stream = torch.cuda.Stream()
        self.warm(warmup)

    def warm(self, n):
        """Eager steps on the capture stream (allocator pools, library handles, workspaces, lazy HIP module loads)."""
        s = self.stream
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(n):
                self.out = self.step_fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        return self

    def capture(self):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=self.stream):
            self.out = self.step_fn()
        self.graph = g
        return self

    def __call__(self):
        if self.graph is None:
            return self.warm(1).out
        self.graph.replay()
        return self.out
