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


class TwoPhaseStep(object):
    """Forward + backward of drn_amd.model.mainModel as TWO hipGraphs that share one memory pool, split where backward
    leaves the trunk: phase 1 = front forward, trunk forward, trunk backward (stops at g0 and the gates of levels 1..);
    phase 2 = backward of the front (prop_fc, position embedding, gate projections, query encoder).  Between the two
    replays the caller launches the all-reduce of the trunk's gradient buckets, which then overlaps phase 2 -- the
    multi-GPU counterpart of overlapping collectives with backward when backward is a replayed graph.

    model: mainModel; batch: its 5 device-resident arguments; loss_of: loss dict -> scalar; reducer: GradReducer built
    with groups=[model.trunk_parameters(), model.front_parameters()]; between(): called after phase 1 (launch
    collectives there)."""

    def __init__(self, model, batch, loss_of, reducer, between=None):
        self.model, self.batch, self.loss_of, self.reducer, self.between = model, batch, loss_of, reducer, between
        self.stream = torch.cuda.Stream()
        self.g1 = self.g2 = None
        self.out = None
        self._carry = None

    def _phase1(self):
        m, (tok, qlen, feats, pse, gt) = self.model, self.batch
        self.reducer.zero()
        g0, gates = m.forward_front(tok, qlen, feats, pse)
        g0d = g0.detach().requires_grad_()
        gd = [gates[0].detach()] + [g.detach().requires_grad_() for g in gates[1:]]
        _, losses = m.forward_trunk(g0d, gd, gt)
        self.loss_of(losses).backward()
        self.reducer.collect(self.reducer.group_buckets[0])
        self._carry = ([g0] + list(gates[1:]), [g0d.grad] + [g.grad for g in gd[1:]])
        return losses

    def _phase2(self):
        outs, grads = self._carry
        torch.autograd.backward(outs, grads)
        self.reducer.collect(self.reducer.group_buckets[1])
        self._carry = None

    def _eager(self):
        self.out = self._phase1()
        if self.between is not None:
            self.between()
        self._phase2()

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
        self.g1, self.g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g1, stream=self.stream):
            self.out = self._phase1()
        with torch.cuda.graph(self.g2, stream=self.stream, pool=self.g1.pool()):
            self._phase2()
        return self

    def __call__(self):
        if self.g1 is None:
            self.warm(1)
            return self.out
        self.g1.replay()
        if self.between is not None:
            self.between()
        self.g2.replay()
        return self.out
