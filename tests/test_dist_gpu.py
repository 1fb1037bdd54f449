"""The N>1 code path of bench.py on ONE GPU: two ranks (one process each, gloo between them, both on cuda:0) run the
three-phase hipGraph step (drn_amd.graph.TwoPhaseStep: trunk / input stage / query side, the trunk's and prop_fc's gradient
exchanges launched between the phases) + the fused clip+Adam for a few steps.  Checked on every rank:
  (i)   the exchanged gradients are the mean of the two ranks' local gradients (recomputed from scratch, plain backward);
  (ii)  after the optimizer steps every parameter is bit-identical on both ranks;
  (iii) graph replay == eager: the same steps without capture give the same parameters.
RCCL itself needs one GPU per rank (the driver's multi-GPU run); everything around the collective is what runs here."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, case):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(rank), DRN_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
        from drn_amd import functional as DF
        from drn_amd.dist import GradReducer, init_from_env
        from drn_amd.graph import TwoPhaseStep
        from drn_amd.model import mainModel
        from drn_amd.optim import FusedAdam
        from drn_amd.utils.synthetic import VOCAB_SIZE, as_namespace, default_cfg, seeded_state_dict, synthetic_batch
        init_from_env(backend="gloo")
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        dtype_name, stage, B, T, D = case
        cdt = torch.bfloat16 if dtype_name == "bf16" else torch.float32
        cfg = default_cfg("C3D" if D == 4096 else "TINY", D, stage)

        def build():
            m = mainModel(VOCAB_SIZE, as_namespace(cfg), compute_dtype=cdt)
            m.load_state_dict(seeded_state_dict(m, 0))
            return m.to(dev).train()

        # same queries on both ranks (token shapes depend on the seed), different clips / ground truth per rank
        base = synthetic_batch(B, T, D, seed=1)
        batches = []
        for r in range(world):
            other = synthetic_batch(B, T, D, seed=11 + r)
            b = list(base)
            b[2], b[4] = other[2], other[4]
            batches.append([x.to(dev) for x in b])
        loss_of = DF.loss_total

        # (i) reference: mean of the two local gradients at the initial weights
        want = None
        for r in range(world):
            mr = build()
            _, ls = mr(*batches[r])
            loss_of(ls).backward()
            g = {k: p.grad.detach().clone() for k, p in mr.named_parameters() if p.grad is not None}
            want = g if want is None else {k: want[k] + g[k] for k in want}
        want = {k: v / world for k, v in want.items()}

        def make(graph):
            m = build()
            params = [p for p in m.parameters() if p.requires_grad]
            # bench.py's N > 1 layout: four parts, the gate projections' bucket exchanged during the query encoder's backward
            red = GradReducer(params, world_size=world, overlap=False, bucket_bytes=1 << 30,
                              groups=[m.trunk_parameters(), m.input_parameters(), m.gate_parameters(), m.encoder_parameters()],
                              adjacent=m.grad_stack_groups())
            opt = FusedAdam(red, lr=1e-3, max_norm=0.5)
            core = TwoPhaseStep(m, batches[rank][:5], loss_of, red, between=lambda k: red.reduce(red.group_buckets[k]))

            def fwd_bwd_exchange():
                red.rearm()
                losses = core()
                assert core.NPHASES == 4
                tm = []
                red.finish(timings=tm)
                assert [i for i, _, _ in tm] == list(range(len(red.buckets)))      # every bucket's wait is timed, in bucket order
                return losses
            return m, red, opt, core, fwd_bwd_exchange

        m, red, opt, core, run = make(graph=True)
        run()                                                   # eager three-phase, gradients exchanged, no optimizer yet
        torch.cuda.synchronize()
        worst = 0.0
        for k, p in m.named_parameters():
            if k in want:
                ref = want[k]
                # (the buckets hold the all-reduced SUM; FusedAdam's kernels apply 1/world themselves: grad_scale)
                err = float((p.grad * opt.grad_scale - ref).abs().max())
                tol = 1e-4 * max(float(ref.abs().max()), 1e-6) + 1e-7
                assert err <= tol, ("mean of local gradients", k, err, tol)
                worst = max(worst, err)
        opt.step()
        run(); opt.step()                                       # second eager step (warm-up for the capture)
        core.capture()
        for _ in range(3):
            run(); opt.step()                                   # three replayed steps
        torch.cuda.synchronize()
        # (iii) the same five steps without capture
        m2, red2, opt2, core2, run2 = make(graph=False)
        for _ in range(5):
            run2(); opt2.step()
        torch.cuda.synchronize()
        for (k, a), (_, b) in zip(m.named_parameters(), m2.named_parameters()):
            assert torch.equal(a, b), ("replay vs eager", k, float((a - b).abs().max()))
        # (ii) identical parameters on both ranks
        mine = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu()
        other = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(other, mine)
        assert all(torch.equal(o, mine) for o in other), "parameters differ between ranks after Adam"
        assert not torch.equal(mine, torch.cat([p.detach().reshape(-1) for p in build().parameters()]).cpu()), "no training happened"
        q.put((rank, "ok"))
        dist.destroy_process_group()
    except Exception as e:                                       # noqa: BLE001 -- report to the parent instead of hanging it
        import traceback
        q.put((rank, "FAILED: %s\n%s" % (e, traceback.format_exc())))


@pytest.mark.parametrize("case", [("f32", 3, 4, 32, 64), ("bf16", 3, 32, 256, 4096)],
                         ids=["f32-B4-T32-D64", "bf16-bench-shape-stage3"])
def test_three_phase_graph_step_two_ranks_one_gpu(case):
    """The second case is BASELINE configs[2] at its per-GPU size (32 clips, T=256, D=4096, third-stage losses) in the benchmarked
    dtype: what each of the 8 ranks runs, here with 2 ranks."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, case)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=900) for _ in procs)
    for p in procs:
        p.join(timeout=120)
    assert got == [(0, "ok"), (1, "ok")], got


def test_bench_eight_ranks_one_gpu():
    """`python bench.py --gpus 8` end to end at world = 8 -- the launch the driver's scaling run makes -- with the eight ranks sharing
    this box's one GPU over gloo (DRN_FORCE_DEVICE=0, DRN_DIST_BACKEND=gloo) on tiny shapes: rank plumbing (torch.distributed.run,
    RANK / LOCAL_RANK / WORLD_SIZE), the four-phase graph step with its bucket order, the max-over-ranks timing and the per-rank /
    per-bucket `exposed_ms` reporting.  Replaces the reference's single-process nn.DataParallel entry (main.py:99).  RCCL itself
    needs one GPU per rank; everything around the collective runs here."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DRN_FORCE_DEVICE="0", DRN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--batch", "2",
                        "--T", "32", "--D", "64", "--stage", "3", "--cpu-steps", "0", "--no-f32", "--no-other-configs", "--no-trainer",
                        "--no-kernel-timing"], cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-4000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                      # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["rccl_ranks"] == 8
    assert d["config"]["global_batch"] == 16 and d["config"]["parallelism"] == "dp8" and "gloo" in d["backend"]
    assert abs(d["value"] - 16 / (d["ms_per_step"] * 1e-3)) <= 0.01 * d["value"]          # whole-job clips/s over the max-over-ranks time
    pr = d["per_rank"]
    assert len(pr["clips_per_s"]) == 8 and len(pr["allreduce_exposed_ms_per_step"]) == 8
    names = pr["exposed_ms_by_bucket"]["buckets"]
    assert len(names) == 4 and names[0].startswith("trunk") and names[1].startswith("input") and names[2].startswith("gate") and \
        names[3].startswith("query"), names              # the order backward finishes the parts in
    by_rank = pr["exposed_ms_by_bucket"]["by_rank"]
    assert len(by_rank) == 8 and all(len(v) == 4 and all(x >= 0.0 for x in v) for v in by_rank)
    assert pr["exposed_ms_by_bucket"]["max_over_ranks"] == [max(v[i] for v in by_rank) for i in range(4)]
