"""More than one foreground channel (fcos_num_class > 2; model/fcos.py:27,43, model/loss.py:149-213, model/inference.py:59-120) -- off
every shipped config of the reference, served by chunked head calls, the fused loss on channel 0 + the class-general focal kernel on the
background-only channels, and the per-level post-processor.  The fp32 path is pinned to outputs recorded from the reference in
tests/test_model_gpu.py (tiny_k2 / tiny_k3 cases); here: a larger shape against the oracle, bf16 against fp32, the captured step against
eager launches, and the trainer's evaluation, which has to take the records path."""
import pytest
import torch

from drn_amd.utils.synthetic import VOCAB_SIZE, as_namespace, default_cfg, seeded_state_dict, synthetic_batch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build(cls, cfg, dev="cpu", **kw):
    m = cls(VOCAB_SIZE, as_namespace(cfg), **kw)
    m.load_state_dict(seeded_state_dict(m, 0))
    return m.to(dev).train()


@pytest.mark.parametrize("K,stage", [(3, 1), (5, 3), (2, 3)])
def test_wide_classifier_matches_the_oracle_at_a_training_shape(K, stage):
    from drn_amd.model import mainModel
    from oracle import drn_oracle as O
    cfg = default_cfg("TINY", 64, stage)
    cfg["fcos_num_class"] = K + 1
    batch = list(synthetic_batch(8, 64, 64, seed=3))
    mo, mh = build(O.mainModel, cfg), build(mainModel, cfg, DEV)
    _, lo = mo(*batch)
    sum(lo.values()).backward()
    hb = [b.to(DEV) for b in batch]
    _, lh = mh(*hb)
    sum(v for v in lh.values()).backward()
    for k in ("loss_cls", "loss_reg", "loss_iou"):
        a, b = float(lh[k].reshape(-1)[0]), float(lo[k].reshape(-1)[0])
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (k, a, b)
    go, gh = dict(mo.named_parameters()), dict(mh.named_parameters())
    # (the head's own parameters tightly; deep in the chain a ReLU input within 1e-7 of zero may fall on the other side in two correct
    # fp32 implementations -- tests/test_model_gpu.py has the account -- and moves an early layer's gradient by a few 1e-3)
    for name, tol in (("fcos.head.cls_logits.weight", 2e-3), ("fcos.head.cls_logits.bias", 2e-3), ("fcos.head.cls_tower.0.weight", 1e-2),
                      ("prop_fc.weight", 1e-2)):
        a, b = gh[name].grad.double().cpu(), go[name].grad.double()
        assert a.shape == b.shape and float((a - b).norm()) <= tol * max(float(b.norm()), 1e-9), (name, float((a - b).norm()), float(b.norm()))
    assert gh["fcos.head.cls_logits.weight"].shape[0] == K


def test_wide_classifier_in_bf16_and_in_a_captured_step():
    from drn_amd.graph import GraphedStep
    from drn_amd.model import mainModel
    cfg = default_cfg("TINY", 64, 1)
    cfg["fcos_num_class"] = 4
    batch = [b.to(DEV) for b in synthetic_batch(8, 64, 64, seed=4)]
    m32 = build(mainModel, cfg, DEV)
    m16 = build(mainModel, cfg, DEV, compute_dtype=torch.bfloat16)
    _, l32 = m32(*batch)
    _, l16 = m16(*batch)
    for k in ("loss_cls", "loss_reg"):
        a, b = float(l16[k].reshape(-1)[0]), float(l32[k].reshape(-1)[0])
        assert abs(a - b) <= 3e-2 * max(1.0, abs(b)), (k, a, b)
    # captured vs eager training (fused Adam, gradient buckets): two models from the same seed, n steps each
    from drn_amd.dist import GradReducer
    from drn_amd.optim import FusedAdam

    def make():
        m = build(mainModel, cfg, DEV, compute_dtype=torch.bfloat16)
        red = GradReducer([p for p in m.parameters() if p.requires_grad], world_size=1)
        opt = FusedAdam(red, lr=1e-3, max_norm=0.5)

        def step():
            red.zero()
            _, losses = m(*batch)
            sum(losses.values()).backward()
            red.finish()
            opt.step()
            return losses
        return m, step, red
    n = 5
    m1, step1, red1 = make()
    eager = [float(step1()["loss_cls"]) for _ in range(n)]
    m2, step2, red2 = make()
    g = GraphedStep(step2, warmup=2).capture()
    replay = [float(g()["loss_cls"]) for _ in range(n - 2)]
    torch.cuda.synchronize()
    assert eager[0] != eager[-1]
    for k, v in enumerate(replay):
        assert abs(v - eager[2 + k]) <= 2e-3 * max(1.0, abs(v)), (k, replay, eager)
    a, b = m1.fcos.head.cls_logits.weight, m2.fcos.head.cls_logits.weight
    assert a.shape[0] == 3 and torch.allclose(a, b, atol=2e-3, rtol=2e-3)
    red1.remove(); red2.remove()


def test_trainer_evaluation_takes_the_records_path():
    from drn_amd.model import mainModel
    from drn_amd import trainer as TR
    import bench as B
    cfg = default_cfg("TINY", 64, 1)
    cfg["fcos_num_class"] = 4
    m = build(mainModel, cfg, DEV)
    bs = [B.collate_like([t.to(DEV) if torch.is_tensor(t) else t for t in synthetic_batch(4, 32, 64, seed=10 + i)], ["v%d_%d" % (i, j) for j in range(4)])
          for i in range(2)]
    tr = TR.Trainer(m, 1, lr=1e-3, clip_gradient=0.5)
    loss, topks, acc, res = tr.evaluate(bs, with_results=False)
    assert loss == loss and len(acc) >= 1 and all(0.0 <= float(a) <= 1.0 for row in (acc if isinstance(acc[0], (list, tuple)) else [acc]) for a in row)
    tr.train_epoch(bs)                                   # ... and a training epoch runs
    if tr.reducer is not None:
        tr.reducer.remove()
