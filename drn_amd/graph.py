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

from . import functional as DF


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


class TwoPhaseStep(object):
    """Forward + backward of drn_amd.model.mainModel as THREE hipGraphs that share one memory pool, split where backward
    leaves one part of the model for the next (the class keeps its first name):
      phase 1  forward of everything + backward of the trunk (backbone, FPN, heads, losses); stops at g0 and the gates;
      phase 2  backward of the input stage (prop_fc, position embedding): its 67 MB weight gradient is the largest;
      phase 3  backward of the gate projections and the query encoder.
    After phase k the caller's `between(k)` launches the all-reduce of that part's gradient bucket, which then overlaps
    the next phase's replay -- the multi-GPU counterpart of overlapping collectives with backward when backward is a
    replayed graph.  Only the query side's bucket (26 MB) is left exposed.

    model: mainModel; batch: its 5 device-resident arguments; loss_of: loss dict -> scalar; reducer: GradReducer built
    with groups=[model.trunk_parameters(), model.input_parameters(), model.query_parameters()]."""

    NPHASES = 3

    def __init__(self, model, batch, loss_of, reducer, between=None):
        self.model, self.batch, self.loss_of, self.reducer, self.between = model, batch, loss_of, reducer, between
        self.stream = torch.cuda.Stream()
        self.graphs = None
        self.out = None
        self._carry = None

    def _phase(self, k):
        m, red = self.model, self.reducer
        if k == 0:
            tok, qlen, feats, pse, gt = self.batch
            red.zero()
            gates = m.encode_query(tok, qlen)
            gd = [g.detach().requires_grad_() for g in gates]
            g0, _ = m.forward_front(tok, qlen, feats, pse, gates=gd)
            g0d = g0.detach().requires_grad_()
            _, losses = m.forward_trunk(g0d, gd, gt)
            DF.backward(self.loss_of(losses))                     # trunk parameters, g0d.grad, gd[1:].grad
            self._carry = (g0, g0d, gates, gd)
            self.out = losses
        elif k == 1:
            g0, g0d, gates, gd = self._carry
            torch.autograd.backward([g0], [g0d.grad])             # prop_fc, position_transform, gd[0].grad
        else:
            g0, g0d, gates, gd = self._carry
            torch.autograd.backward(list(gates), [g.grad for g in gd])     # gate projections, query encoder
            self._carry = None
        red.collect(red.group_buckets[k])

    def _eager(self):
        for k in range(self.NPHASES):
            self._phase(k)
            if self.between is not None and k + 1 < self.NPHASES:
                self.between(k)

    def warm(self, n):
        s = self.stream
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(n):
                self._eager()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        return self

    def capture(self):
        self.graphs = []
        for k in range(self.NPHASES):
            g = torch.cuda.CUDAGraph()
            kw = {} if k == 0 else {"pool": self.graphs[0].pool()}
            with torch.cuda.graph(g, stream=self.stream, **kw):
                self._phase(k)
            self.graphs.append(g)
        return self

    def __call__(self):
        if self.graphs is None:
            self.warm(1)
            return self.out
        for k, g in enumerate(self.graphs):
            g.replay()
            if self.between is not None and k + 1 < self.NPHASES:
                self.between(k)
        return self.out
