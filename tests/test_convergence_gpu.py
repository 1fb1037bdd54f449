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
STEPS, EVAL_CLIPS = 400, 8192     # 8192 held-out clips: one clip = 0.012 pt, binomial sigma at 99.5 % = 0.08 pt (1024 clips: 0.22 pt -- the 0.3 pt band was 1.4 sigma)


def run(dtype, jitter=0.0):
    from drn_amd import trainer as TR
    from drn_amd.model import mainModel
    from drn_amd.utils.synthetic import VOCAB_SIZE, as_namespace, default_cfg, planted_batches, seeded_state_dict
    torch.manual_seed(0)
    m = mainModel(VOCAB_SIZE, as_namespace(default_cfg("TINY", D, 1)), compute_dtype=dtype)
    m.load_state_dict(seeded_state_dict(m, 0))
    if jitter:                                  # a second f32 run from weights moved by 1e-6 relative: how far two runs of the
        g = torch.Generator().manual_seed(7)     # SAME arithmetic drift apart at this learning rate (trajectory chaos, not dtype)
        with torch.no_grad():
            for p in m.parameters():
                p.mul_(1.0 + jitter * torch.randn(p.shape, generator=g))
    m = m.to("cuda:0")
    tr = TR.Trainer(m, 1, lr=1e-3, clip_gradient=0.5, graph=True)
    train = planted_batches(STEPS, B, T_PROPS, D, seed=1, noise=NOISE)
    test = planted_batches(EVAL_CLIPS // B, B, T_PROPS, D, seed=2, noise=NOISE)
    losses = []
    for i in range(0, STEPS, 50):
        losses.append(tr.train_epoch(train[i:i + 50]))          # mean summed loss of 50 steps
    val_loss, topks, accs, _ = tr.evaluate(test, iou_topk={"iou": [0.5, 0.7], "topk": [1, 5]})     # main.py:362 uses IoU 0.5
    tr.reducer.remove()
    return np.array(losses), val_loss, [100.0 * a for a in accs]


def test_bf16_trains_like_f32_on_a_planted_signal():
    l32, v32, a32 = run(torch.float32)
    l16, v16, a16 = run(torch.bfloat16)
    l3j, v3j, a3j = run(torch.float32, jitter=1e-6)
    fmt = "%s: loss per 50 steps %s | val %.4f | IoU 0.5: R@1 %.2f R@5 %.2f | IoU 0.7: R@1 %.2f R@5 %.2f"
    print(fmt % ("f32   ", np.round(l32, 4).tolist(), v32, a32[0], a32[1], a32[2], a32[3]))
    print(fmt % ("bf16  ", np.round(l16, 4).tolist(), v16, a16[0], a16[1], a16[2], a16[3]))
    print(fmt % ("f32+1e-6", np.round(l3j, 4).tolist(), v3j, a3j[0], a3j[1], a3j[2], a3j[3]))
    # both learn the task ...
    assert l32[-1] < 0.5 * l32[0] and l16[-1] < 0.5 * l16[0], (l32, l16)
    assert a32[0] >= R1_FLOOR and a16[0] >= R1_FLOOR, (a32, a16)
    # ... along the same curve: mean loss of every 50-step window within the band, widened by how far the second f32 run (started
    # 1e-6 away) is from the first in that window -- in the steep part of the curve (steps 50-150) two runs of the SAME arithmetic
    # are 3-20 % apart, and which of them is ahead changes with every last-bit change in any kernel
    assert np.all(np.abs(l16 - l32) <= LOSS_BAND * np.maximum(l32, 0.05) + 2.0 * np.abs(l3j - l32)), (l32, l16, l3j)
    # ... to the same accuracy on the reference's metric (R@1 / R@5 at IoU 0.5, main.py:362): north_star's 0.3 pt
    # (two f32 runs started 1e-6 apart end 0.0-0.3 pt apart themselves after 400 steps at lr 1e-3: the band is taken around that spread)
    for j in (0, 1):
        assert abs(a16[j] - a32[j]) <= R1_BAND + abs(a3j[j] - a32[j]), (j, a32, a16, a3j)
    # the stricter IoU 0.7 numbers are still moving after 400 steps at lr 1e-3 and differ between two f32 runs that start 1e-6
    # apart; bf16 must not be further from f32 than that spread plus a margin
    for j in (2, 3):
        assert abs(a16[j] - a32[j]) <= abs(a3j[j] - a32[j]) + IOU07_MARGIN, (j, a32, a16, a3j)


NOISE = 1.5
R1_FLOOR, LOSS_BAND, R1_BAND, IOU07_MARGIN = 60.0, 0.15, 0.3, 10.0     # measured values: DESIGN.md section 4 (round 3)
