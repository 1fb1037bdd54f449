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
import contextlib
import gc

import torch

from . import functional as DF


@contextlib.contextmanager
def capture_graph(g, stream, **kw):
    """torch.cuda.graph(g, stream, capture_error_mode="thread_local") with Python's cyclic garbage collector switched OFF for the
    duration: a collection that happens to run in the middle of a capture may finalise an unreachable CUDAGraph / stream / event
    of some earlier, already dropped step object (anything that sat in a reference cycle) -- destroying those while a capture is
    open aborts the process (seen as a rare "Fatal Python error: Aborted ... Garbage-collecting" in the trainer tests).
    thread_local: a hipHostMalloc from another thread (a DataLoader's pin_memory thread) must not fail the capture."""
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local", **kw):
            yield
    finally:
        if was:
            gc.enable()


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
        with capture_graph(g, self.stream):
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
        """A reducer built with FOUR groups -- [trunk, input stage, model.gate_parameters(), model.encoder_parameters()] -- splits
        the last phase in two (round 5): phase 3 = backward of the gate projections (20 MB of gradients), phase 4 = backward of
        the query encoder (25 MB); the projections' bucket then travels while the query encoder's backward (~0.15 ms of BiLSTM
        steps) replays, and only the encoder's own bucket is left exposed.  Same kernels on the same values as the three-phase
        split (the projections run as their own autograd node, DF.gate_projections)."""
        self.model, self.batch, self.loss_of, self.reducer, self.between = model, batch, loss_of, reducer, between
        self.NPHASES = 4 if len(getattr(reducer, "group_buckets", [])) == 4 else 3
        self.stream = torch.cuda.Stream()
        self.graphs = None
        self.out = None
        self._carry = None

    def _phase(self, k):
        m, red = self.model, self.reducer
        if k == 0:
            tok, qlen, feats, pse, gt = self.batch[:5]
            red.zero()
            cmds = cd = None
            if self.NPHASES == 4:
                cmds = m.encode_commands(tok, qlen)
                cd = [c.detach().requires_grad_() for c in cmds]
                gates = m.project_gates(cd)
            else:
                gates = m.encode_query(tok, qlen)
            gd = [g.detach().requires_grad_() for g in gates]
            g0, _ = m.forward_front(tok, qlen, feats, pse, gates=gd)
            g0d = g0.detach().requires_grad_()
            g0d._drn_tail = getattr(g0, "_drn_tail", None)        # (conv0's backward produces the position-embedding gradients)
            _, losses = m.forward_trunk(g0d, gd, gt)
            DF.backward(self.loss_of(losses))                     # trunk parameters, g0d.grad, gd[1:].grad
            self._carry = (g0, g0d, gates, gd, cmds, cd)
            self.out = losses
        elif k == 1:
            g0, g0d, gates, gd, cmds, cd = self._carry
            torch.autograd.backward([g0], [g0d.grad])             # prop_fc, position_transform, gd[0].grad
        elif k == 2:
            g0, g0d, gates, gd, cmds, cd = self._carry
            torch.autograd.backward(list(gates), [g.grad for g in gd])     # gate projections (, query encoder: three phases)
            if self.NPHASES == 3:
                self._carry = None
        else:
            g0, g0d, gates, gd, cmds, cd = self._carry
            torch.autograd.backward(list(cmds), [c.grad for c in cd])      # query encoder
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
            with capture_graph(g, self.stream, **kw):
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


class DualStreamStep(object):
    """One training step of drn_amd.model.mainModel on ONE GPU as seven linear hipGraphs on two streams, so that the query
    side -- ~35 small, latency-bound launches each way that use a fraction of the chip -- runs BESIDE bandwidth- / MFMA-bound
    work it does not depend on (a single hipGraph with parallel branches leaves ROCm 7.2's fast submission path, see
    DESIGN.md; separate linear graphs joined by events do not):

        main stream                                              side stream
        prep     cast / transpose of the features, weight warm   q_fwd   query encoder + gate projections
        trunk    prop_fc .. losses, backward down to the gates   (waits for q_fwd)
                 (weight gradients deferred, functional._defer)
        wgrads   the deferred weight gradients                   q_bwd   gate projections + query encoder backward
        opt      clip + Adam + fp32 weight copies                (waits for q_bwd)
        repack   bf16 weight copies                              q_fwd of the NEXT step starts after `opt`

    Results are bit-identical to the single-stream step: deferral only reorders independent launches and every kernel is
    deterministic (tests/test_graph_gpu.py).  model: mainModel; batch: its 5 device-resident arguments; loss_of: loss dict ->
    scalar; reducer: GradReducer over the trainable parameters (steal mode); opt: FusedAdam on that reducer."""

    PHASES = ("q_fwd", "prep", "trunk", "q_bwd", "wgrads", "opt", "repack")
    SIDE = ("q_fwd", "q_bwd")

    def __init__(self, model, batch, loss_of, reducer, opt, wgrads_first=True, side_priority=0, main_first=False):
        self.model, self.batch, self.loss_of, self.reducer, self.opt = model, batch, loss_of, reducer, opt
        self.main_first = main_first
        # the side stream's launches are few workgroups each and latency-bound: at high priority they get CU slots ahead of the
        # main stream's thousands of queued workgroups instead of waiting for a whole wave of them to drain
        self.main, self.side = torch.cuda.Stream(), torch.cuda.Stream(priority=side_priority)
        self.wgrads_first = wgrads_first       # small weight gradients before the prop_fc one (it owns every CU while it runs)
        self.graphs = None
        self.out = None
        self._c = {}
        self._ev = dict((k, torch.cuda.Event()) for k in ("q_fwd", "trunk", "q_bwd", "opt", "start"))
        self._fresh = True
        have = set(id(p) for p in reducer.params)
        self._qparams = [p for p in model.query_parameters() if id(p) in have]

    # ---- the phases (each runs on the stream its row in the table says)
    def _phase(self, name):
        self._run_phase(name)

    def _run_phase(self, name):
        m, red, c = self.model, self.reducer, self._c
        tok, qlen, feats, pse, gt = self.batch[:5]
        if name == "q_fwd":
            c["gates"] = m.encode_query(tok, qlen)
        elif name == "prep":
            from . import ops
            ops.CAST_THROTTLE, was = getattr(self, "prep_throttle", 0), ops.CAST_THROTTLE
            try:
                c["prep"] = m.prepare_input(feats, pse)
            finally:
                ops.CAST_THROTTLE = was
        elif name == "trunk":
            red.zero()
            gd = c["gd"] = [g.detach().requires_grad_() for g in c["gates"]]
            DF.begin_defer()
            try:
                g0, _ = m.forward_front(tok, qlen, feats, pse, gates=gd, prep=c.pop("prep"))
                _, losses = m.forward_trunk(g0, gd, gt)
                DF.backward(self.loss_of(losses))            # trunk + input stage; the gates' gradients land in gd[i].grad
            finally:
                c["jobs"] = DF.take_deferred()
            self.out = losses
        elif name == "q_bwd":
            grads = torch.autograd.grad(c["gates"], self._qparams, [g.grad for g in c["gd"]], allow_unused=True)
            red.adopt(self._qparams, grads)
        elif name == "wgrads":
            jobs = c.pop("jobs")
            if not self.wgrads_first:
                jobs = jobs[::-1]
            for job in jobs:
                job()
        elif name == "opt":
            red.finish()
            self.opt.step(repack=False)
            self.opt.repack(codes=(0,))                      # ops.F32: the query side's stacks, needed by the next q_fwd
        else:
            self.opt.repack(codes=(1,))                      # ops.BF16

    def _schedule(self, run, M, Q):
        """Issue the seven phases with their cross-stream dependencies; run(name) launches phase `name` on the current stream."""
        ev = self._ev
        if self._fresh:                       # first step on these streams: the side stream starts behind everything queued so far
            ev["start"].record(M)
            Q.wait_event(ev["start"])
            self._fresh = False
        else:
            Q.wait_event(ev["opt"])           # the previous step's parameters
        if self.main_first:
            with torch.cuda.stream(M):
                run("prep")
            with torch.cuda.stream(Q):
                run("q_fwd")
                ev["q_fwd"].record(Q)
            with torch.cuda.stream(M):
                M.wait_event(ev["q_fwd"])
                run("trunk")
                ev["trunk"].record(M)
                run("wgrads")
            with torch.cuda.stream(Q):
                Q.wait_event(ev["trunk"])
                run("q_bwd")
                ev["q_bwd"].record(Q)
            with torch.cuda.stream(M):
                M.wait_event(ev["q_bwd"])
                run("opt")
                ev["opt"].record(M)
                run("repack")
            return
        with torch.cuda.stream(Q):
            run("q_fwd")
            ev["q_fwd"].record(Q)
        with torch.cuda.stream(M):
            run("prep")
            M.wait_event(ev["q_fwd"])
            run("trunk")
            ev["trunk"].record(M)
        with torch.cuda.stream(Q):
            Q.wait_event(ev["trunk"])
            run("q_bwd")
            ev["q_bwd"].record(Q)
        with torch.cuda.stream(M):
            run("wgrads")
            M.wait_event(ev["q_bwd"])
            run("opt")
            ev["opt"].record(M)
            run("repack")

    def warm(self, n):
        """Eager two-stream steps on the capture streams (see GraphedStep.warm for why these streams and no others)."""
        cur = torch.cuda.current_stream()
        self.main.wait_stream(cur)
        self._fresh = True
        for _ in range(n):
            self._schedule(self._phase, self.main, self.side)
        cur.wait_stream(self.main)
        cur.wait_stream(self.side)
        torch.cuda.synchronize()
        return self

    def capture(self):
        graphs, pools = {}, {}
        for name in self.PHASES:
            side = name in self.SIDE
            g = torch.cuda.CUDAGraph()
            kw = {"pool": pools[side]} if side in pools else {}
            with capture_graph(g, self.side if side else self.main, **kw):
                self._phase(name)
            pools.setdefault(side, g.pool())
            graphs[name] = g
        self._c.clear()                       # (the graphs' private pools keep every block they captured)
        self.graphs = graphs
        self._fresh = True
        return self

    def __call__(self):
        if self.graphs is None:
            self.warm(1)
            return self.out
        M = torch.cuda.current_stream()
        if getattr(self, "_last_main", None) != M.cuda_stream:
            self._fresh, self._last_main = True, M.cuda_stream
        self._schedule(lambda name: self.graphs[name].replay(), M, self.side)
        return self.out


class ForkedStep(DualStreamStep):
    """The DualStreamStep schedule as ONE hipGraph with two branches: the query encoder beside the input preparation, the query
    side's backward beside the weight gradients.  Round 2 gave up on branches ("host 0.05 -> 1.9 ms per replay"); measured in
    round 4 on ROCm 7.2 the host side of a replay is 0.3-0.4 ms, all of it hidden behind the ~2 ms the device needs, and the
    step 2.18-2.21 -> 2.11-2.15 ms (scripts/experiments/one_graph_two_branches.py).  The second branch must sit on a stream of
    the SAME priority as the first: captured on a `priority=-1` stream (what DualStreamStep asked for) three candidates in four
    ran the step in 4.1-4.4 ms, and the high-priority hardware queues left behind slowed every later two-stream user of the
    process (the trainer's H2D look-ahead 12.9 -> 9.0 k clips/s); from the default pool all candidates measure the same.
    Schedule variants, all on working streams: small weight gradients first or the prop_fc one first, within 1 %;
    `split_gate=True` (the prop_fc GEMM un-gated so that the query encoder can run beside IT, the gate as a pass of its own) within
    1 % too -- two BiLSTM steps next to the 512-workgroup GEMM take 100-130 us instead of 7 (rocprofv3 timeline); a THIRD branch
    (every weight gradient launched the moment backward has its operands, beside the data-gradient chain) 2.26 ms, slower than the
    linear graph: the MFMA kernels of two branches only take each other's CUs.  Results equal the single-stream step bit for bit
    (tests/test_graph_gpu.py)."""

    def __init__(self, model, batch, loss_of, reducer, opt, split_gate=False, main=None, side=None, rotate=None):
        """main / side: the streams to capture on (default: new ones).  `main` must be the stream every earlier step of this
        model ran on (autograd's AccumulateGrad nodes are bound to it).
        rotate (default False: an explicit opt-in -- it measured slower, and between calls the parameters lag one update, which a
        caller who checkpoints or evaluates without flush() would read; needs a reducer that keeps the query side's gradients in
        buckets of their own, i.e. built with groups=[model.query_parameters(), the rest]): OPTIMIZER-FIRST order.  A call applies the optimizer to the gradients the
        PREVIOUS call left in the reducer's buckets and then runs forward + backward of its own batch:

            main   norm -> Adam(query side) -> Adam(rest) + weight copies -> input prep -> (join) trunk fwd/bwd -> weight gradients
            side                            +-> query encoder forward ---------------------+              +-> query side backward

        so the query encoder's forward -- ~16 dependent, latency-bound launches, 130 us alone -- runs beside the ~200 us of
        bandwidth-bound optimizer kernels, whose short-lived workgroups leave it room, instead of in front of the prop_fc GEMM,
        which had to wait ~100 us for the gate (profiles/r05_*_forked_timeline.txt).  The sequence of operations over a run is the
        one of the plain loop -- f0 b0 | o0 f1 b1 | o1 f2 b2 | ... | o_n -- bit for bit: the first call only computes gradients
        (`prime`), `flush()` applies the pending update; read parameters (evaluation, checkpoints) only after a flush."""
        model.split_gate = bool(split_gate)
        self._streams = (main, side)
        qp = set(id(p) for p in model.query_parameters())
        gb = getattr(reducer, "group_buckets", [])
        can = len(gb) >= 2 and len(gb[0]) >= 1 and all(id(p) in qp for b in gb[0] for p in b.params) and \
            not any(id(p) in qp for g in gb[1:] for b in g for p in b.params)
        if rotate and not can:
            raise ValueError("ForkedStep(rotate=True) needs GradReducer(groups=[model.query_parameters(), the other parameters])")
        self.rotate = bool(rotate)
        self._primed = False
        import os
        if os.environ.get("DRN_FORK_ROTATE") is not None:              # experiment switches (scripts/experiments/ab_r05_c.sh)
            self.rotate = can and os.environ["DRN_FORK_ROTATE"] == "1"
        self._env_wf = os.environ.get("DRN_FORK_WGRADS_FIRST")
        self._env_mf = os.environ.get("DRN_FORK_MAIN_FIRST")
        # (side stream from the DEFAULT-priority pool: high-priority streams bring a second set of hardware queues into being, and
        # with them around everything else in the process that uses two streams ran slower afterwards -- the trainer's H2D
        # look-ahead 12.9 -> 9.0 k clips/s, evaluation 22 -> 15 k, measured in bench.py after a capture on priority -1 streams)
        super(ForkedStep, self).__init__(model, batch, loss_of, reducer, opt, wgrads_first=False, side_priority=0)
        # the input cast runs beside the query encoder's forward, which is the critical path there (prop_fc waits for the gate): at
        # full rate it stretches the latency-bound query launches 2-3 x (first dense product 18 -> 47 us, an LSTM step 7.5 -> 20);
        # held to 128 resident workgroups it takes 121 instead of 64 us -- still hidden -- and the query forward 164 instead of
        # 174 us: 2.028 / 2.019 -> 2.012 / 2.013 ms per step (64: 2.055, 96: 2.012-2.024, 192: 2.010-2.027, 256: 2.016-2.020)
        # (re-swept at the end of round 5, non-temporal cast loads and the position embedding beside it: 64: 1.96, 96: 1.92, 128: 1.912,
        # 192: 1.905, 256: 1.907, unthrottled 1.927 ms -- two rounds each in one box)
        self.prep_throttle = int(os.environ.get("DRN_FORK_PREP_THROTTLE", "192"))
        if self._env_wf is not None:
            self.wgrads_first = self._env_wf == "1"
        if self._env_mf is not None:
            self.main_first = self._env_mf == "1"
        if self._streams[0] is not None:
            self.main = self._streams[0]
        if self._streams[1] is not None:
            self.side = self._streams[1]
        self.graph = None

    # ---- optimizer-first order
    def _run_phase(self, name):
        red = self.reducer
        if name == "norm":
            self.opt.norm()
        elif name == "opt_q":
            self.opt.update(red.group_buckets[0])
            self.opt.repack(codes=(0,), buckets=red.group_buckets[0])          # ops.F32: the query side's stacks
        elif name == "opt_rest":
            rest = [b for g in red.group_buckets[1:] for b in g]
            self.opt.update(rest)
            self.opt.repack(buckets=rest)
        elif name == "collect":
            red.finish()                                     # (one process: gradients produced outside their sinks are moved in)
        else:
            super(ForkedStep, self)._run_phase(name)

    def _schedule(self, run, M, Q):
        if not self.rotate:
            return super(ForkedStep, self)._schedule(run, M, Q)
        ev = self._ev
        ev.setdefault("optq", torch.cuda.Event())
        if self._fresh:                       # first step on these streams: the side stream starts behind everything queued so far
            ev["start"].record(M)
            Q.wait_event(ev["start"])
            self._fresh = False
        if self.main_first:                   # issue order: the main branch's launches ahead of the side branch's wherever both are ready
            with torch.cuda.stream(M):
                if self._primed:
                    run("norm")
                    run("opt_q")
                    ev["optq"].record(M)
                    run("opt_rest")
                run("prep")
            with torch.cuda.stream(Q):
                if self._primed:
                    Q.wait_event(ev["optq"])
                run("q_fwd")
                ev["q_fwd"].record(Q)
            with torch.cuda.stream(M):
                M.wait_event(ev["q_fwd"])
                run("trunk")
                ev["trunk"].record(M)
                run("wgrads")
            with torch.cuda.stream(Q):
                Q.wait_event(ev["trunk"])
                run("q_bwd")
                ev["q_bwd"].record(Q)
            with torch.cuda.stream(M):
                M.wait_event(ev["q_bwd"])
                run("collect")
            self._primed = True
            return
        with torch.cuda.stream(M):
            if self._primed:                  # the update the previous call's gradients are waiting for
                run("norm")
                run("opt_q")
                ev["optq"].record(M)
        with torch.cuda.stream(Q):
            if self._primed:
                Q.wait_event(ev["optq"])
            run("q_fwd")
            ev["q_fwd"].record(Q)
        with torch.cuda.stream(M):
            if self._primed:
                run("opt_rest")
            run("prep")
            M.wait_event(ev["q_fwd"])
            run("trunk")
            ev["trunk"].record(M)
        with torch.cuda.stream(Q):
            Q.wait_event(ev["trunk"])
            run("q_bwd")
            ev["q_bwd"].record(Q)
        with torch.cuda.stream(M):
            run("wgrads")
            M.wait_event(ev["q_bwd"])
            run("collect")
        self._primed = True

    def flush(self):
        """Apply the update the last call's gradients are waiting for (optimizer-first order; a no-op otherwise and when nothing
        is pending).  Eager launches on the main stream; the next call primes again."""
        if not (self.rotate and self._primed):
            return
        cur = torch.cuda.current_stream()
        self.main.wait_stream(cur)
        self.main.wait_stream(self.side)
        with torch.cuda.stream(self.main):
            for name in ("norm", "opt_q", "opt_rest"):
                self._phase(name)
        cur.wait_stream(self.main)
        self._primed = False
        self._fresh = True

    def _capture_once(self, pool=None):
        if self.rotate and not self._primed:
            self.warm(1)                                     # the captured step starts with an update: there must be gradients
        g = torch.cuda.CUDAGraph()
        self._fresh = True
        with capture_graph(g, self.main, **({"pool": pool} if pool is not None else {})):
            self._schedule(self._phase, self.main, self.side)
            self.main.wait_stream(self.side)
        self._c.clear()
        return g

    def capture(self, tries=2, probe=6, pool=None):
        """Capture (with `tries` > 1: on that many different side streams, each candidate replayed `probe` times -- real training
        steps, counted in `tuning_steps` -- and the fastest graph kept: a safety net from the time one stream in four worked;
        probe=0 takes the first candidate without running anything).  The main stream never changes: autograd's AccumulateGrad
        nodes are bound to it."""
        import time
        log = self.__dict__.setdefault("probe_log", [])
        self.tuning_steps = 0                    # training steps executed in here (timing replays, warm-up of candidate streams)
        best, worst = None, 0.0
        for i in range(max(int(tries), 1)):
            if i:
                self.side = torch.cuda.Stream()
                self.warm(1)                     # (allocator pools / workspaces of the new stream)
            if i:
                self.tuning_steps += 1
            g = self._capture_once(pool)
            if probe <= 0:                       # no timing (tests): the first candidate is taken, no training step is spent
                best = (0.0, g, self.side, self.out)
                break
            g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(probe):
                g.replay()
            torch.cuda.synchronize()
            self.tuning_steps += 1 + probe
            t = (time.perf_counter() - t0) / probe
            log.append(round(t * 1e3, 3))
            worst = max(worst, t)
            if best is None or t < best[0]:
                best = (t, g, self.side, self.out)       # (the losses a replay returns live in THAT capture's memory)
            else:
                del g
            if i and best[0] < 0.75 * worst:
                break
        self.probe_ms = best[0] * 1e3
        self.graph, self.side, self.out = best[1], best[2], best[3]
        return self

    def __call__(self):
        if self.graph is None or (self.rotate and not self._primed):       # (after a flush(): the replay starts with an update)
            self.warm(1)
            return self.out
        self.graph.replay()
        return self.out
