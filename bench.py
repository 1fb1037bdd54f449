#!/usr/bin/env python
"""DRN hot-path benchmark on MI355X (driver contract: one JSON line from rank 0).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): Charades-STA-shaped synthetic C3D features, T=256 proposals, D=4096,
batch 32 per GPU, first-stage losses, bf16 storage / fp32 accumulation.  One step = forward + backward of
drn_amd.model.mainModel + gradient all-reduce (N>1) + clip_grad_norm_(0.5) + Adam (main.py:218-243), inputs
already resident in HBM.  `value` = clips/s over all ranks (weak scaling: per-GPU batch fixed).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from drn_amd.utils.synthetic import VOCAB_SIZE, as_namespace, default_cfg, seeded_state_dict, synthetic_batch  # noqa: E402

PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}      # MI355X dense MFMA peaks (MI355X_MICROARCH.md)
PEAK_HBM_TBS = 8.0


def path_flops(T, D, stage):
    """Algorithmic FLOPs per clip of the path (2 x multiply-adds of every contraction, query encoder excluded):
    forward by component, and what one training step executes = 3 x forward - the prop_fc input gradient (the features need
    no gradient, as in the reference's autograd) - in stage 1 the backward of the frozen IoU branch."""
    SL = T + T // 2 + T // 4
    f = {"prop_fc": 2.0 * T * D * D, "conv0": 2.0 * T * (D + 256) * 3 * 256,
         "conv1": 2.0 * (T // 2) * 256 * 3 * 512, "conv2": 2.0 * (T // 4) * 512 * 3 * 1024,
         "fpn": 2.0 * (T * 256 + (T // 2) * 512 + (T // 4) * 1024) * 512 + 2.0 * SL * 512 * 3 * 512,
         "towers": 2.0 * SL * 512 * 3 * 1024, "outs": 2.0 * SL * 512 * 3 * 3,
         "iou_branch": 2.0 * SL * (1024 * 512 + 512 * 3 * 256 + 256)}
    fwd = sum(f.values())
    step = 3.0 * fwd - f["prop_fc"] - (2.0 * f["iou_branch"] if stage == 1 else 0.0)
    return {"fwd": fwd, "step": step, "fpn_heads_fwd": f["fpn"] + f["towers"] + f["outs"] + f["iou_branch"]}


def time_graph(fn, reps=20):
    """Median-free mean time (ms) of `reps` replays of a hipGraph of fn(), HIP events on the replay stream."""
    from drn_amd.graph import GraphedStep
    g = GraphedStep(fn, warmup=2).capture()
    g()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def other_config_lines(dev, reps=10):
    """Every other BASELINE.json config at the size ONE GPU runs it, each as its own hipGraph of the whole step (forward + backward
    + clip + Adam), `reps` replays timed with HIP events: configs[2] and [4] at their 8-way data-parallel per-GPU batch (256 / 8,
    128 / 8), configs[3] whole; configs[4] in bf16 and in f32 (its tolerance sweep).  path_frac as in `roofline.path`."""
    from drn_amd import dist as ddist
    from drn_amd import functional as DF
    from drn_amd.model import mainModel
    from drn_amd.optim import FusedAdam
    lines = {}
    todo = [("configs[2] per-GPU: T=256 D=4096 B=32 stage 3 (batch 256 over 8 GPUs)", 32, 256, 4096, 3, "bf16"),
            ("configs[3]: T=512 D=1024 B=64 stage 1 (I3D-shaped bandwidth probe)", 64, 512, 1024, 1, "bf16"),
            ("configs[4] per-GPU: T=1024 D=500 B=16 stage 1 (batch 128 over 8 GPUs)", 16, 1024, 500, 1, "bf16"),
            ("configs[4] per-GPU, f32: T=1024 D=500 B=16 stage 1", 16, 1024, 500, 1, "f32")]
    for name, B, T, D, stage, dt in todo:
        try:
            cdt = torch.bfloat16 if dt == "bf16" else torch.float32
            cfg = default_cfg("C3D" if D == 4096 else "SYN", D, stage)
            m = build(mainModel, cfg, dev, compute_dtype=cdt)
            params = stage_params(m, stage)
            m.train()
            red = ddist.GradReducer(params, world_size=1, overlap=True, adjacent=m.grad_stack_groups(), bucket_bytes=1 << 30)
            opt = FusedAdam(red, lr=1e-3, max_norm=0.5)
            batch = [b.to(dev) for b in synthetic_batch(B, T, D, seed=7)]

            def step():
                red.zero()
                _, ls = m(*batch)
                DF.backward(DF.loss_total(ls))
                red.finish()
                opt.step()
                return ls
            t = time_graph(step, reps=reps)
            fl = path_flops(T, D, stage)
            lines[name] = {"B": B, "T": T, "D": D, "stage": stage, "dtype": dt, "ms_per_step": round(t, 3),
                           "clips_per_s": round(B / (t * 1e-3), 1),
                           "path_frac": round(fl["step"] * B / (t * 1e-3) / 1e12 / PEAK_TFLOPS[dt], 4)}
            red.remove()
            del m, opt, red, batch
            torch.cuda.empty_cache()
        except Exception as e:
            lines[name] = {"error": "%s: %s" % (type(e).__name__, str(e).split(chr(10))[0])}
    return lines


def build(model_cls, cfg, device, **kw):
    m = model_cls(VOCAB_SIZE, as_namespace(cfg), **kw)
    m.load_state_dict(seeded_state_dict(m, 0))
    return m.to(device)


def stage_params(model, stage):
    """main.py:124-138: stage 1 freezes iou_scores / mix_fc."""
    for n, p in model.named_parameters():
        if stage == 1 and ("iou_scores" in n or "mix_fc" in n):
            p.requires_grad_(False)
    # (mainModel.learned_parameters drops QueryEncoder.textualAttention, which the reference declares and never calls: its
    # gradients stay None there and torch's clip_grad_norm_ / Adam skip such parameters, main.py:140,238-243)
    return model.learned_parameters() if hasattr(model, "learned_parameters") else [p for p in model.parameters() if p.requires_grad]


def cpu_baseline(cfg, B, T, D, stage, steps):
    """The CPU oracle (oracle/drn_oracle.py, pinned to the reference's goldens) timed on this box's host cores."""
    from oracle import drn_oracle as O
    m = build(O.mainModel, cfg, "cpu")
    stage_params(m, stage)
    m.train()
    batch = synthetic_batch(B, T, D, seed=1)
    times = []
    for i in range(steps + 1):
        t0 = time.perf_counter()
        m.zero_grad(set_to_none=True)
        _, losses = m(*batch)
        sum(l for l in losses.values()).backward()
        times.append(time.perf_counter() - t0)
    t = sorted(times[1:])[len(times[1:]) // 2]
    return {"value": B / t, "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "median of %d fwd+bwd steps (after 1 warm-up) of B=%d,T=%d,D=%d stage-%d, fp32, oracle/drn_oracle.py"
                      % (steps, B, T, D, stage)}


def collate_like(batch, names):
    """synthetic_batch's 7 model arguments -> the 8-tuple drn_amd.data.collate_data yields (dataset.py:180-224)."""
    tok, qlen, feats, pse, gt, nprops, nframes = batch
    return (names, pse, feats, gt, tok, qlen, nprops, nframes)


def trainer_lines(cfg, dev, cdt, stage, B, D, Ts, steps, graph_modes=(True, False)):
    """Throughput of the TRAINING LOOP a user runs (train.py -> drn_amd.trainer.Trainer.train_epoch, main.py:198-252), not of a
    hand-built step: a fresh model per line, eight device-resident synthetic batches with different query lengths, `steps` steps
    through train_epoch after the warm-up epoch that captures the hipGraphs.  T=256 is the benchmarked shape, T=32 the number of
    proposals Charades-STA really has (model/loss.py:98)."""
    from drn_amd import trainer as TR
    from drn_amd.model import mainModel
    out = {}
    for T in Ts:
        batches = [collate_like([t.to(dev) if torch.is_tensor(t) else t for t in synthetic_batch(B, T, D, seed=100 + i)], ["v%d" % i] * B)
                   for i in range(8)]
        for graph in graph_modes:
            m = build(mainModel, cfg, dev, compute_dtype=cdt)
            tr = TR.Trainer(m, stage, lr=1e-3, clip_gradient=0.5, graph=graph)
            tr.train_epoch(batches)                          # warm-up: every geometry passes warm-up steps
            tr.train_epoch(batches)                          # ... and capture
            torch.cuda.synchronize()
            # ONE epoch of `steps` steps (the eight batches in turn): train_epoch returns the epoch's mean loss, i.e. ends with a host
            # sync, and a real epoch is hundreds of steps (Charades-STA: 388 at batch 32) -- eight-step epochs, as timed before round 5,
            # charged every step an eighth of that pipeline bubble
            n_ep = max(1, steps // len(batches))
            epoch = batches * n_ep
            t0 = time.perf_counter()
            tr.train_epoch(epoch)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            key = "T%d_%s" % (T, "graph" if graph else "eager")
            out[key] = {"clips_per_s": round(B * len(batches) * n_ep / dt, 1), "ms_per_step": round(dt / (len(batches) * n_ep) * 1e3, 3),
                        "graphs": len([s for s in tr._slots.values() if s.graph is not None]) if graph else 0}
            if graph:
                # the same loop fed from PINNED HOST batches (what a DataLoader hands over): the H2D copy of the features rides on
                # a copy stream one batch ahead -- the PCIe-inclusive rate, never the headline value.  train.py's hand-over for a
                # bf16 model is bf16 features (drn_amd.data.collate_data(feature_dtype=...): rounded in the DataLoader workers by
                # the rule the step's cast applies, B x T x D x 2 bytes per step); `_fp32` = the reference's fp32 hand-over
                for key, fdt in (("T%d_graph_host_inputs" % T, cdt), ("T%d_graph_host_inputs_fp32" % T, torch.float32)):
                    if key.endswith("_fp32") and cdt == torch.float32:
                        continue
                    hb = [tuple((t.to(fdt) if i == 2 else t).cpu().pin_memory() if torch.is_tensor(t) else t for i, t in enumerate(b))
                          for b in batches]
                    tr.train_epoch(hb)
                    tr.train_epoch(hb)
                    tr.train_epoch(hb)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    tr.train_epoch(hb * n_ep)
                    torch.cuda.synchronize()
                    dth = time.perf_counter() - t0
                    out[key] = {"clips_per_s": round(B * len(hb) * n_ep / dth, 1), "ms_per_step": round(dth / (len(hb) * n_ep) * 1e3, 3),
                                "h2d_MB_per_step": round(B * T * D * (2 if fdt == torch.bfloat16 else 4) / 1e6, 1)}
                    del hb
            if graph and T == Ts[-1]:
                # evaluation loop (main.py:270-366) as Trainer.fit runs it: eval-mode forward, post-processor, Recall@k with
                # temporal NMS on the device (drn_eval_recall), one host copy at the end; `records` = the path that also builds the
                # reference's raw-results records and runs the host evaluator on them (Trainer.evaluate's default)
                ev = batches[:4] * 4
                for key, kw in (("evaluate_T%d" % T, {"with_results": False}), ("evaluate_T%d_records" % T, {})):
                    tr.evaluate(ev, **kw)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    tr.evaluate(ev, **kw)
                    out[key] = {"clips_per_s": round(B * len(ev) / (time.perf_counter() - t0), 1)}
                out["evaluate_T%d" % T]["note"] = ("Trainer.evaluate(with_results=False), what fit() runs: eval forward + drn_postprocess + "
                                                   "drn_eval_recall (temporal NMS, R@1/R@5 on the device); *_records: + the host-side "
                                                   "raw-results records and the host evaluator")
            if tr.reducer is not None:
                tr.reducer.remove()
            del m, tr
    out["note"] = ("Trainer.train_epoch (what train.py runs) on 8 device-resident synthetic batches, B=%d, query lengths 3..8 padded to "
                   "multiples of 4; graph = hipGraph replay per input geometry (Trainer(graph=True), train.py's default), eager = every "
                   "kernel launched from Python; *_host_inputs = the graph loop fed from pinned host batches, features in the compute "
                   "dtype as train.py's DataLoader hands them over (H2D copy one batch ahead on a copy stream: the PCIe-inclusive "
                   "rate); *_host_inputs_fp32 = the same with fp32 features (the reference's hand-over)" % B)
    return out


def launch_plan(gpus, argv, env, device_count):
    """What `python bench.py --gpus N` has to do before anything else (replaces the reference's single-process
    nn.DataParallel entry, main.py:99): None = run in this process (N = 1, or already one rank of a torch.distributed.run
    launch); a command list = re-exec under torch.distributed.run with one rank per GPU over RCCL.  Raises SystemExit when the
    box has fewer than N GPUs -- an N = 1 number must never be printed for an N > 1 request.  DRN_FORCE_DEVICE (test mode:
    several ranks share one device over gloo) waives the device-count check."""
    if gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" in env:
        if int(env["WORLD_SIZE"]) != gpus:
            raise SystemExit("--gpus %d but WORLD_SIZE=%s" % (gpus, env["WORLD_SIZE"]))
        return None
    if gpus == 1:
        return None
    if env.get("DRN_FORCE_DEVICE") is None and device_count < gpus:
        raise SystemExit("bench.py --gpus %d: this box has %d GPU(s); refusing to print a smaller run's number" % (gpus, device_count))
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps (100 x 2.2 ms: the ~0.4 ms of barrier + first-replay latency "
                                                             "around the timed region weighs 0.2 %, not the 1 % it does at 20)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU")
    ap.add_argument("--T", type=int, default=256)
    ap.add_argument("--D", type=int, default=4096)
    ap.add_argument("--stage", type=int, default=1)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--cpu-steps", type=int, default=12, help="timed oracle steps for cpu_baseline (~1 s each on 32 threads); 0 disables")
    ap.add_argument("--no-f32", dest="f32_line", action="store_false",
                    help="skip the extra exact-f32 (1e-4 parity mode) timing of the same workload reported as `f32`")
    ap.add_argument("--verbose", action="store_true", help="per-kernel MFMA timing table on stderr")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-graph", dest="graph", action="store_false", help="launch every kernel from Python instead of replaying a hipGraph")
    ap.add_argument("--dual-stream", action="store_true",
                    help="N=1: replay the step as seven linear hipGraphs on two streams (drn_amd.graph.DualStreamStep: query side beside "
                         "input prep / deferred weight gradients) instead of ONE linear hipGraph; measured 0.5 %% SLOWER on ROCm 7.2 "
                         "(DESIGN.md section 5), kept as an experiment")
    ap.add_argument("--no-forked", dest="forked", action="store_false",
                    help="N=1: never use the two-branch hipGraph (by default it replaces the linear one when a short A/B at warm-up "
                         "time says it is faster)")
    ap.add_argument("--torch-adam", action="store_true", help="torch clip_grad_norm_ + optim.Adam instead of the fused HIP step")
    ap.add_argument("--dump-gemms", default=None, metavar="PATH",
                    help="write the MFMA launches of one step in launch order [(tag, flops)] as JSON (scripts/gemm_table.py joins them "
                         "with a rocprofv3 kernel trace of the replayed graph)")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE", help="drn_tune(KEY, VALUE) before anything runs (experiments)")
    ap.add_argument("--no-other-configs", dest="other_configs", action="store_false",
                    help="skip `other_configs` (BASELINE configs[2..4] at their single-GPU size, ~10 graph-replayed steps each)")
    ap.add_argument("--no-trainer", dest="trainer_line", action="store_false",
                    help="skip the `trainer` object (clips/s through drn_amd.trainer.Trainer.train_epoch at T=256 and T=32, graph and eager)")
    args = ap.parse_args()
    cmd = launch_plan(args.gpus, sys.argv[1:], os.environ, torch.cuda.device_count() if torch.cuda.is_available() else 0)
    if cmd is not None:
        # one process per GPU: the plain command launches its own ranks (a driver that mirrors the N = 1 command gets N ranks)
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        os.execvpe(cmd[0], cmd, env)

    def note(msg):                       # progress on stderr with --verbose (where a multi-rank run stopped, if it did)
        if args.verbose:
            print("[bench rank %s] %s" % (os.environ.get("RANK", "0"), msg), file=sys.stderr, flush=True)

    from drn_amd import dist as ddist
    from drn_amd import functional as DF
    from drn_amd import ops
    from drn_amd.model import mainModel
    # DRN_DIST_BACKEND=gloo + DRN_FORCE_DEVICE=0 lets several ranks share one GPU to exercise the N>1 code path on a
    # single-GPU box (test only; the real runs use RCCL, one GPU per rank)
    for kv in args.tune:
        from drn_amd._lib import check, lib
        k, v = kv.split("=")
        check(lib().drn_tune(k.encode(), int(v)), "drn_tune")
    rank, local, world = ddist.init_from_env(backend=os.environ.get("DRN_DIST_BACKEND"))
    if os.environ.get("DRN_FORCE_DEVICE") is not None:
        local = int(os.environ["DRN_FORCE_DEVICE"])
    if world != args.gpus:
        raise SystemExit("--gpus %d but %d rank(s) initialised" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    B, T, D, stage = args.batch, args.T, args.D, args.stage
    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    cfg = default_cfg("C3D" if D == 4096 else "SYN", D, stage)

    model = build(mainModel, cfg, dev, compute_dtype=cdt)
    params = stage_params(model, stage)
    model.train()
    # N>1 with hipGraph: forward+backward replay as THREE graphs split where backward leaves one part of the model
    # (drn_amd.graph.TwoPhaseStep): the trunk's gradients are all-reduced while the input stage's backward replays, prop_fc's
    # 67 MB while the query side's backward replays, the query side's 26 MB afterwards.  One bucket per part: xGMI rings are
    # per-link bound, so few large messages.
    # Eager mode overlaps 32 MB bucket all-reduces with backward from post-accumulate-grad hooks.
    deferred = args.graph and world > 1
    if deferred:
        # four parts, one bucket each, in the order backward finishes them: trunk (41 MB), input stage (prop_fc: 67 MB), gate
        # projections (20 MB), query encoder (25 MB) -- each travels while the next part's backward replays; only the last is
        # exposed by construction (round 5: the query side's 45 MB used to be one exposed bucket)
        reducer = ddist.GradReducer(params, world_size=world, overlap=False, bucket_bytes=1 << 30,
                                    groups=[model.trunk_parameters(), model.input_parameters(), model.gate_parameters(),
                                            model.encoder_parameters()],
                                    adjacent=model.grad_stack_groups())
    else:
        # one process: nothing to overlap, so ONE bucket (one norm pass + one Adam launch instead of one pair per 32 MB)
        reducer = ddist.GradReducer(params, world_size=world, overlap=True, adjacent=model.grad_stack_groups(),
                                    **({"bucket_bytes": 1 << 30} if world == 1 else {}))
    if args.torch_adam:
        opt = torch.optim.Adam(params, lr=1e-3)                      # main.py:140
    else:
        from drn_amd.optim import FusedAdam
        opt = FusedAdam(reducer, lr=1e-3, max_norm=0.5)              # clip_grad_norm_(0.5) + Adam in two HIP kernels/bucket
    batch = [b.to(dev) for b in synthetic_batch(B, T, D, seed=1 + rank)]     # everything resident in HBM, lengths included

    loss_of = lambda losses: losses["loss_iou"] if stage == 2 else DF.loss_total(losses)             # main.py:222-225

    with torch.no_grad():            # the untrained model's loss on the resident batch (checked against the trained one after the run)
        initial_total = float(loss_of(model(*batch)[1]).detach().float().reshape(-1)[0])

    ar_events = None                 # N>1: HIP events around the exchange wait of every timed step (exposed all-reduce time)
    ar_bucket_events = []            # ... and around each bucket's own wait (bucket index, start, end)

    def opt_step():
        if ar_events is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            reducer.finish(timings=ar_bucket_events)
            e1.record()
            ar_events.append((e0, e1))
        else:
            reducer.finish()
        if args.torch_adam:
            torch.nn.utils.clip_grad_norm_(params, 0.5)              # main.py:238-239
        opt.step()

    def step():
        reducer.zero()
        _, losses = model(*batch)
        DF.backward(loss_of(losses))
        opt_step()
        return losses

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    run, mode, degraded, launch_ab = step, "eager", False, None
    if args.graph:
        # all warm-up steps run on the capture stream (see drn_amd/graph.py), then the step is captured once
        from drn_amd.graph import GraphedStep
        try:
            if world == 1 and args.dual_stream and not args.torch_adam:
                # seven linear hipGraphs on two streams: the query side (small latency-bound launches) runs beside the
                # input preparation forward and beside the deferred weight gradients backward
                from drn_amd.graph import DualStreamStep
                run = DualStreamStep(model, batch[:5], loss_of, reducer, opt).warm(max(args.warmup, 2)).capture()
                mode = "hipGraph replay of the full step: 7 linear graphs on 2 streams (query side beside input prep / deferred weight gradients)"
            elif world == 1:
                # the whole step (query encoder, HIP path forward+backward, fused clip+Adam) replays as ONE hipGraph: the linear
                # one, or -- when it measures faster in a short A/B of the two captures here, during warm-up -- the same launches
                # with two branches (drn_amd.graph.ForkedStep: query side beside input preparation / weight gradients; same bits)
                run = GraphedStep(step, warmup=max(args.warmup, 2)).capture()
                mode = "hipGraph replay of the full step"
                if args.forked and not args.torch_adam:
                    try:
                        # (its own model / reducer / optimizer: autograd's AccumulateGrad nodes remember the stream they were created
                        # on, so two captures of one model on different streams race -- drn_amd/graph.py)
                        from drn_amd.graph import ForkedStep
                        from drn_amd.optim import FusedAdam as _FA
                        model_f = build(mainModel, cfg, dev, compute_dtype=cdt)
                        params_f = stage_params(model_f, stage)
                        model_f.train()
                        # (ONE bucket.  With the query side's gradients in a bucket of their own ForkedStep runs optimizer-first -- the
                        # query encoder's forward beside the Adam kernels of the pending update; built, bit-identical, measured in
                        # round 5: 2.020 ms against 2.003 here -- the two optimizer launches per kernel cost 36 us, and next to the
                        # bandwidth-bound Adam kernels every latency-bound query launch runs 3-5 x longer: the overlap bought 13 us)
                        gkw = {}
                        if os.environ.get("DRN_BENCH_FORK_ROTATE") == "1":      # (experiment: the optimizer-first order needs the query side's own bucket)
                            qp = set(id(p) for p in model_f.query_parameters())
                            gkw = {"groups": [[p for p in params_f if id(p) in qp], [p for p in params_f if id(p) not in qp]]}
                        reducer_f = ddist.GradReducer(params_f, world_size=1, overlap=True, adjacent=model_f.grad_stack_groups(),
                                                      bucket_bytes=1 << 30, **gkw)
                        forked = ForkedStep(model_f, batch[:5], loss_of, reducer_f, _FA(reducer_f, lr=1e-3, max_norm=0.5), rotate=bool(gkw)).warm(
                            max(args.warmup, 2)).capture()

                        def probe(fn, n=12):
                            fn(); fn()
                            torch.cuda.synchronize()
                            tp = time.perf_counter()
                            for _ in range(n):
                                fn()
                            torch.cuda.synchronize()
                            return (time.perf_counter() - tp) / n * 1e3
                        ab = [(probe(run), probe(forked)) for _ in range(2)]
                        t_lin, t_fork = min(a for a, _ in ab), min(b for _, b in ab)
                        launch_ab = {"linear_ms": round(t_lin, 3), "forked_ms": round(t_fork, 3)}
                        if t_fork < 0.995 * t_lin:
                            run = forked
                            mode = ("hipGraph replay of the full step, two branches, optimizer-first order (query encoder forward beside the "
                                    "Adam kernels of the pending update, its backward beside the weight gradients; same operation sequence as "
                                    "the plain loop)") if forked.rotate else \
                                "hipGraph replay of the full step, two branches (query side beside input prep / weight gradients)"
                        else:
                            reducer_f.remove()
                            del forked, model_f, reducer_f
                    except Exception as e:
                        print("forked graph not used (%s: %s)" % (type(e).__name__, str(e).split(chr(10))[0]), file=sys.stderr)
            else:
                # forward+backward replay as three hipGraphs per rank; RCCL all-reduces + fused optimizer stay outside them
                from drn_amd.graph import TwoPhaseStep
                core = TwoPhaseStep(model, batch[:5], loss_of, reducer,
                                    between=lambda k: reducer.reduce(reducer.group_buckets[k]))

                def run():
                    reducer.rearm()                   # hooks only run eagerly / at capture time
                    losses = core()
                    opt_step()                        # reduces the query side's bucket, waits for all, clip + Adam
                    return losses
                for _ in range(max(args.warmup, 2)):
                    run()
                core.capture()
                mode = ("hipGraph replay of forward+backward in %d phases; each part's all-reduce (trunk / prop_fc / gate projections) "
                        "overlaps the next phase; optimizer eager" % core.NPHASES)
            run()
        except Exception as e:                                          # keep the eager path measurable
            print("hipGraph capture failed (%s: %s); running eager" % (type(e).__name__, str(e).split(chr(10))[0]), file=sys.stderr)
            torch.cuda.synchronize()
            reducer.overlap = True
            run, mode, degraded = step, "eager (capture failed)", True
    else:
        for _ in range(args.warmup):
            step()
    note("warm-up / capture done (%s)" % mode)
    barrier()
    if world > 1:
        ar_events = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = run()
    barrier()
    dt = time.perf_counter() - t0
    note("timed loop done")
    ops.check_watchdogs()            # an in-launch exchange that gave up waiting lets invalid values through: such a run is not a measurement
    # ... and neither is a run that did not train: the step fits ONE resident batch, so its loss must have stayed finite and not grown.
    # (Round 6: a build with a store-data hazard in one epilogue ran 3 % FASTER -- the corrupted activations drove the loss to a constant
    # 6.28, the degenerate operands toggled fewer bits, the chip drew less power and clocked every MFMA kernel 3-7 % higher;
    # profiles/HISTORY.md.  The parity suite caught it; this keeps such a number out of a bench line as well.)
    final_total = float(loss_of(losses).detach().float().reshape(-1)[0])
    if not (final_total == final_total and abs(final_total) != float("inf")) or final_total > 2.0 * initial_total + 1e-3:
        raise SystemExit("bench.py: the loss went from %.4g to %.4g over the run: the step is not training, not a measurement"
                         % (initial_total, final_total))
    exposed_ms = sum(a.elapsed_time(b) for a, b in ar_events) / max(len(ar_events), 1) if ar_events else 0.0
    exposed_by_bucket = {}
    for i, a, b in ar_bucket_events:
        exposed_by_bucket[i] = exposed_by_bucket.get(i, 0.0) + a.elapsed_time(b)
    n_timed = max(len(ar_events), 1) if ar_events else 1
    exposed_by_bucket = [round(exposed_by_bucket.get(i, 0.0) / n_timed, 4) for i in range(len(reducer.buckets))]
    ar_events = None
    # the same step launched kernel by kernel from Python (what the hipGraph replay saves): wall clock over a few steps
    eager_ms = None
    if args.graph and world == 1 and not args.no_kernel_timing:
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        te = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        eager_ms = (time.perf_counter() - te) / 10 * 1e3
    # per-kernel timing of the MFMA GEMMs for the roofline object: HIP events around each launch, on the launch
    # stream, over a few extra eager steps of the same workload (events cannot sit inside a replayed graph)
    timers = None
    if not args.no_kernel_timing:
        timers = ops.kernel_timer = []
        for _ in range(max(3, min(args.steps, 5))):
            step()
        torch.cuda.synchronize()
        ops.kernel_timer = None
        note("kernel timing done")
        timed_steps = max(3, min(args.steps, 5))
        if args.dump_gemms and rank == 0:
            per = len(timers) // timed_steps
            with open(args.dump_gemms, "w") as f:
                json.dump([[t[0], t[1]] for t in timers[-per:]], f)
    per_rank = None
    if world > 1:
        mine = torch.tensor([dt, exposed_ms] + exposed_by_bucket, device=dev, dtype=torch.float64)
        allr = [torch.empty_like(mine) for _ in range(world)]
        torch.distributed.all_gather(allr, mine)
        per_rank = {"clips_per_s": [round(B * args.steps / float(t[0]), 1) for t in allr],
                    "allreduce_exposed_ms_per_step": [round(float(t[1]), 4) for t in allr],
                    "exposed_ms_by_bucket": {"buckets": ["%s (%.0f MB)" % (nm, b.flat.numel() * 4 / 1e6) for nm, b in
                                                         zip(("trunk", "input stage", "gate projections", "query encoder")
                                                             if deferred else ["bucket %d" % i for i in range(len(reducer.buckets))],
                                                             reducer.buckets)],
                                             "max_over_ranks": [round(max(float(t[2 + i]) for t in allr), 4) for i in range(len(reducer.buckets))],
                                             "by_rank": [[round(float(t[2 + i]), 4) for i in range(len(reducer.buckets))] for t in allr]},
                    "note": "exposed = GPU time the step's stream waits in GradReducer.finish(): per bucket, HIP events around that "
                            "bucket's wait (the exchanges are launched after the phase that produces them and overlap the next phase's "
                            "replay; the last bucket has nothing to hide behind)"}
        dt = max(float(t[0]) for t in allr)
        note("per-rank gather done")
    ms = dt / args.steps * 1e3
    value = B * world * args.steps / dt
    fl = path_flops(T, D, stage)

    # FPN + heads forward (north_star's named region) as its own hipGraph on the backbone outputs of this workload
    fpn_heads = None
    if rank == 0 and args.graph and not args.no_kernel_timing:
        try:
            with torch.no_grad():
                g0, gates = model.forward_front(*batch[:4])
                bb = model.backbone_net.forward_from_stage(g0, gates)

                def fpn_heads_fwd():
                    model.fcos.head.forward_nlc(model.fpn.forward_nlc(bb))
                    DF.flush_bn_counters()
                t_fh = time_graph(fpn_heads_fwd)
            gf = fl["fpn_heads_fwd"] * B / 1e9
            fpn_heads = {"GFLOP": round(gf, 1), "us": round(t_fh * 1e3, 1), "TFLOP/s": round(gf / t_fh, 1),
                         "frac": round(gf / t_fh / PEAK_TFLOPS[args.dtype], 4),
                         "note": "fpn.forward_nlc + fcos.head.forward_nlc (laterals, level convs, towers, cls/reg heads, mix_fc, IoU "
                                 "head, train-mode BN) replayed as one hipGraph on this workload's C1..C3, HIP events over 20 replays"}
        except Exception as e:
            print("fpn+heads timing failed: %s" % e, file=sys.stderr)

    # the same workload in the exact-f32 mode (the reference's own arithmetic, the <=1e-4 parity mode)
    f32_line = None
    if rank == 0 and world == 1 and args.f32_line and args.dtype == "bf16" and args.graph and not args.torch_adam:
        try:
            from drn_amd.optim import FusedAdam
            m32 = build(mainModel, cfg, dev, compute_dtype=torch.float32)
            p32 = stage_params(m32, stage)
            m32.train()
            r32 = ddist.GradReducer(p32, world_size=1, overlap=True, adjacent=m32.grad_stack_groups(), bucket_bytes=1 << 30)
            o32 = FusedAdam(r32, lr=1e-3, max_norm=0.5)

            def step32():
                r32.zero()
                _, ls = m32(*batch)
                DF.backward(loss_of(ls))
                r32.finish()
                o32.step()
                return ls
            t32 = time_graph(step32, reps=min(args.steps, 10))
            f32_line = {"value": round(B / (t32 * 1e-3), 2), "unit": "clips/s", "ms_per_step": round(t32, 3), "dtype": "f32",
                        "path_frac": round(fl["step"] * B / (t32 * 1e-3) / 1e12 / PEAK_TFLOPS["f32"], 4),
                        "note": "same step with compute_dtype=float32 (v_mfma_f32_16x16x4_f32, 157.3 TFLOP/s dense peak): the mode "
                                "tests/test_parity_grad_gpu.py holds to 1e-4 against the oracle"}
            r32.remove()
            del m32, o32, r32
        except Exception as e:
            print("f32 timing failed: %s" % e, file=sys.stderr)

    # the same step fed with features ALREADY in the compute dtype (what train.py's DataLoader hands a bf16 model): no cast pass,
    # only the K-major copy for the prop_fc weight gradient is produced -- the headline's fp32 hand-over (the reference's dtype) costs
    # the cast_transpose pass on top.  Linear graph on both sides (`launch_ab.linear_ms` is the fp32-input twin).
    bf16_feat = None
    if rank == 0 and world == 1 and args.dtype == "bf16" and args.graph and not args.torch_adam and not args.no_kernel_timing:
        try:
            from drn_amd.optim import FusedAdam
            mb = build(mainModel, cfg, dev, compute_dtype=cdt)
            pb = stage_params(mb, stage)
            mb.train()
            rb = ddist.GradReducer(pb, world_size=1, overlap=True, adjacent=mb.grad_stack_groups(), bucket_bytes=1 << 30)
            ob = FusedAdam(rb, lr=1e-3, max_norm=0.5)
            batch_b = list(batch)
            batch_b[2] = batch[2].to(cdt)

            def step_b(feed):
                def fn():
                    rb.zero()
                    _, ls = mb(*feed)
                    DF.backward(loss_of(ls))
                    rb.finish()
                    ob.step()
                    return ls
                return fn
            tb = time_graph(step_b(batch_b), reps=min(args.steps, 20))
            tf = time_graph(step_b(batch), reps=min(args.steps, 20))
            bf16_feat = {"ms_per_step": round(tb, 3), "clips_per_s": round(B / (tb * 1e-3), 1), "fp32_features_ms_per_step": round(tf, 3),
                         "handover_cost_us": round((tf - tb) * 1e3, 1),
                         "note": "one linear hipGraph each, same model: features handed over in bf16 (B x T x D x 2 bytes, rounded by the "
                                 "rule the step's cast applies) vs in fp32 (the headline's and the reference's hand-over); the difference "
                                 "is the cast half of cast_transpose_kernel, a hand-over cost, not kernel time of the path"}
            rb.remove()
            del mb, ob, rb, batch_b
        except Exception as e:
            print("bf16-features timing failed: %s" % e, file=sys.stderr)

    roof = None
    if timers:
        agg = {}
        for tag, flops, e0, e1 in timers:
            a = agg.setdefault(tag, [0.0, 0, flops])
            a[0] += e0.elapsed_time(e1)
            a[1] += 1
        if args.verbose and rank == 0:
            for t_, (ms_, n_, fl_) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
                print("%-60s %3d/step  %8.3f ms/launch  %7.1f TFLOP/s" % (t_, n_ // timed_steps, ms_ / n_, fl_ / (ms_ / n_ * 1e-3) / 1e12),
                      file=sys.stderr)
        tag, (tot_ms, n, flops) = max(agg.items(), key=lambda kv: kv[1][0])
        avg_ms = tot_ms / n
        achieved = flops / (avg_ms * 1e-3) / 1e12
        gemm_ms = sum(a[0] for a in agg.values()) / timed_steps
        # HBM/fabric bytes per launch of that kernel: not measurable from inside the process; taken from the committed
        # rocprofv3 --pmc passes of the same launch (profiles/pmc_traffic.json) when there is one, else null
        traffic, traffic_note = None, None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                ent = json.load(f).get(tag)
            if ent:
                traffic = ent["read_bytes"] + ent["write_bytes"]
                traffic_note = "bytes/launch from profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes); %s" % ent["served_by"]
        except (OSError, ValueError, KeyError):
            pass
        # the same launch INSIDE the replayed graph (what the timed steps run): from the committed rocprofv3 kernel trace of this
        # command (profiles/gemm_in_graph.json, written by scripts/prof_round.sh -> scripts/gemm_table.py), when there is one
        in_graph = None
        try:
            with open(os.path.join(ROOT, "profiles", "gemm_in_graph.json")) as f:
                ent = json.load(f).get(tag)
            if ent:
                in_graph = {"us": ent["us"], "achieved": ent["TFLOP/s"], "frac": ent["frac"],
                            "source": "profiles/gemm_in_graph.json (rocprofv3 --kernel-trace of the replayed step, scripts/gemm_table.py)"}
        except (OSError, ValueError, KeyError):
            pass
        roof = {"bound": "mfma", "kernel": tag, "achieved": round(achieved, 1), "peak": PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_TFLOPS[args.dtype], 4), "in_graph": in_graph, "traffic": traffic, "traffic_note": traffic_note,
                "avg_launch_ms_eager": round(avg_ms, 4),
                "launches_per_step": n // timed_steps, "mfma_kernels_ms_per_step_eager": round(gemm_ms, 3),
                "note": "dominant kernel by total time; flops = 2*M*N*K of that launch; `*_eager` figures (and `achieved`, `frac`) are "
                        "HIP events around each launch of extra EAGER steps on the launch stream (events cannot sit inside a replayed "
                        "graph); inside the replayed graph the same launches run 5-7 % shorter: profiles/*_step_kernel_sequence.txt, "
                        "profiles/*_gemm_table.txt"}
        if args.dtype == "bf16":
            # `peak` is the datasheet clock.  The chip clocks to its power budget: what it SUSTAINS on bf16 MFMA with operands that toggle
            # like data, measured here, now, by a register-only MFMA loop at the issue floor (drn_diag_mfma_sustained; after the timed
            # region).  No scheduling gets a GEMM past this number; the zero-operand figure shows it is the data, not the loop.
            try:
                ops.mfma_sustained(2000)
                sus, sus0 = ops.mfma_sustained(20000), ops.mfma_sustained(20000, zero_operands=True)
                roof["sustained"] = {"achieved_over_sustained": round(achieved / sus["tflops"], 4),
                                     "random_bf16_operands": {k: round(v, 2) for k, v in sus.items()},
                                     "zero_operands": {k: round(v, 2) for k, v in sus0.items()},
                                     "note": "256 workgroups x 4 waves of v_mfma_f32_32x32x16_bf16 on registers only, 32 cycles per MFMA and SIMD "
                                             "= the issue floor; TFLOP/s at the clock the power budget allows (shader cycles / wall time)"}
            except Exception as e:          # (measurement only: never costs the bench line)
                roof["sustained"] = {"error": str(e)}
    # t_bound / t_measured with t_bound = max(FLOPs / peak_mfma, compulsory bytes / peak_hbm) over the WHOLE step
    step_flops = fl["step"] * B * world
    bytes_in = 4.0 * B * world * T * D                       # the feature tensor, read once (fp32 in HBM)
    t_mfma, t_hbm = step_flops / (PEAK_TFLOPS[args.dtype] * 1e12 * world), bytes_in / (PEAK_HBM_TBS * 1e12 * world)
    path = {"TFLOP_per_step": round(step_flops / 1e12, 4), "TFLOP/s": round(step_flops / (ms * 1e-3) / 1e12, 1),
            "path_frac": round(max(t_mfma, t_hbm) / (ms * 1e-3), 4), "bound": "mfma" if t_mfma >= t_hbm else "hbm",
            "compulsory_input_bytes": int(bytes_in),
            "note": "executed FLOPs = 3 x forward - prop_fc input gradient (- frozen IoU branch backward in stage 1), SURVEY 8d table"}
    if roof is not None:
        roof["path"] = path
        roof["path_frac"] = path["path_frac"]
        roof["fpn_heads_fwd"] = fpn_heads
    else:
        roof = {"bound": "mfma", "path": path, "path_frac": path["path_frac"], "fpn_heads_fwd": fpn_heads}

    out = {"metric": "clips/sec fwd+bwd (BxT=256x4096 C3D feats)", "value": round(value, 2), "unit": "clips/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": "Charades-STA-shaped C3D features, T=%d, D=%d, batch %d/GPU, stage-%d losses; "
                                  "step = fwd+bwd+grad all-reduce+clip(0.5)+Adam" % (T, D, B, stage),
                      "global_batch": B * world, "T": T, "D": D, "parallelism": "dp%d" % world, "launch": mode, "launch_ab": launch_ab,
                      "loss_cls": float(losses["loss_cls"].detach().reshape(-1)[0])},
           "roofline": roof}
    out["rccl_ranks"] = torch.distributed.get_world_size() if world > 1 else 1
    out["backend"] = torch.distributed.get_backend() if world > 1 else "none (single process)"
    if degraded:
        out["degraded"] = True          # hipGraph capture failed: `value` is the ~2x slower eager launch mode, see stderr
    if eager_ms is not None:
        out["eager_ms_per_step"] = round(eager_ms, 3)
    if per_rank is not None:
        out["per_rank"] = per_rank
    if f32_line is not None:
        out["f32"] = f32_line
    if bf16_feat is not None:
        out["bf16_features"] = bf16_feat
    if rank == 0 and world == 1 and args.other_configs and args.graph and not args.torch_adam:
        try:
            out["other_configs"] = other_config_lines(dev)
        except Exception as e:
            print("other_configs timing failed: %s: %s" % (type(e).__name__, e), file=sys.stderr)
    if rank == 0 and world == 1 and args.trainer_line and args.graph and not args.torch_adam:
        try:
            del run
            forked = model_f = reducer_f = params_f = None        # (the two-branch capture, its model and its streams)
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            out["trainer"] = trainer_lines(cfg, dev, cdt, stage, B, D, (32, T) if T != 32 else (T,), min(args.steps * 2, 48))
        except Exception as e:
            print("trainer timing failed: %s: %s" % (type(e).__name__, e), file=sys.stderr)
    if rank == 0:
        if world == 1 and args.cpu_steps > 0:
            # stock PyTorch oversubscribes badly on these small convs beyond ~32 threads
            torch.set_num_threads(min(32, os.cpu_count() or 1))
            out["cpu_baseline"] = cpu_baseline(cfg, B, T, D, stage, args.cpu_steps)
        print(json.dumps(out), flush=True)
    note("result printed")
    if world > 1:
        torch.distributed.destroy_process_group()
    note("exit")


if __name__ == "__main__":
    main()
