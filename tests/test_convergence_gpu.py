"""Does the benchmarked dtype TRAIN like the reference's arithmetic?  (missing #3.)

north_star asks for R@1 (IoU 0.5) within 0.3 pt of the reference on Charades-STA; the features are not available here, so the
obtainable evidence is relative: the same planted-signal task (drn_amd.utils.synthetic.planted_batches: the features carry the
ground-truth segment) trained by drn_amd.trainer.Trainer -- the hipGraph loop train.py runs -- once with the exact-f32 kernels
(the mode held to 1e-4 against the reference) and once in bf16 (the benchmarked mode), from the same initial weights on the same
batches, then evaluated on the same held-out clips with the reference's own metric path (drn_amd.metrics = utils/evaluate_utils.py:
score sort, temporal NMS, R@1 / R@5 at IoU 0.5, main.py:362)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

T_PROPS, D, B = 32, 64, 32          # Charades-STA's 32 proposals (model/loss.py:98), a small feature dim, the reference's batch size
STEPS, EVAL_CLIPS = 400, 1024


def run(dtype):
    from drn_amd import trainer as TR
    from drn_amd.model import mainModel
    from drn_amd.utils.synthetic import VOCAB_SIZE, as_namespace, default_cfg, planted_batches, seeded_state_dict
    torch.manual_seed(0)
    m = mainModel(VOCAB_SIZE, as_namespace(default_cfg("TINY", D, 1)), compute_dtype=dtype)
    m.load_state_dict(seeded_state_dict(m, 0))
    m = m.to("cuda:0")
    tr = TR.Trainer(m, 1, lr=1e-3, clip_gradient=0.5, graph=True)
    train = planted_batches(STEPS, B, T_PROPS, D, seed=1)
    test = planted_batches(EVAL_CLIPS // B, B, T_PROPS, D, seed=2)
    losses = []
    for i in range(0, STEPS, 50):
        losses.append(tr.train_epoch(train[i:i + 50]))          # mean summed loss of 50 steps
    val_loss, topks, accs, _ = tr.evaluate(test)
    tr.reducer.remove()
    return np.array(losses), val_loss, 100.0 * accs[0], 100.0 * accs[1]


def test_bf16_trains_like_f32_on_a_planted_signal():
    l32, v32, r1_32, r5_32 = run(torch.float32)
    l16, v16, r1_16, r5_16 = run(torch.bfloat16)
    print("f32 : loss per 50 steps %s | val %.4f | R@1 %.2f R@5 %.2f" % (np.round(l32, 4).tolist(), v32, r1_32, r5_32))
    print("bf16: loss per 50 steps %s | val %.4f | R@1 %.2f R@5 %.2f" % (np.round(l16, 4).tolist(), v16, r1_16, r5_16))
    # both learn the task ...
    assert l32[-1] < 0.5 * l32[0] and l16[-1] < 0.5 * l16[0], (l32, l16)
    assert r1_32 >= R1_FLOOR and r1_16 >= R1_FLOOR, (r1_32, r1_16)
    # ... along the same curve (mean loss of every 50-step window within the band) and to the same accuracy
    assert np.all(np.abs(l16 - l32) <= LOSS_BAND * np.maximum(l32, 0.05)), (l32, l16)
    assert abs(r1_16 - r1_32) <= R1_BAND and abs(r5_16 - r5_32) <= R1_BAND, (r1_32, r1_16, r5_32, r5_16)


R1_FLOOR, LOSS_BAND, R1_BAND = 60.0, 0.15, 1.0          # measured values: DESIGN.md section 4 (round 3)
