"""GPU parity of the device-length BiLSTM (drn_lstm_step_fwd/bwd) against nn.LSTM on packed sequences (fp64 CPU),
i.e. exactly what the reference's QueryEncoder runs (model/language_module.py:38-45)."""
import pytest
import torch
import torch.nn as nn
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,L,E,H,lens", [(5, 7, 300, 512, [7, 6, 4, 2, 1]), (32, 8, 300, 512, None), (3, 4, 20, 64, [4, 4, 1])])
def test_bilstm_matches_packed_nn_lstm(B, L, E, H, lens):
    from drn_amd import functional as DF
    torch.manual_seed(0)
    if lens is None:
        lens = sorted(torch.randint(1, L + 1, (B,)).tolist(), reverse=True)
        lens[0] = L
    lengths = torch.tensor(lens, dtype=torch.int64)
    ref = nn.LSTM(E, H, 1, batch_first=True, bidirectional=True).double()
    x = torch.randn(B, L, E, dtype=torch.float64)
    for b in range(B):
        x[b, lens[b]:] = 0
    xr = x.clone().requires_grad_()
    out, _ = ref(pack_padded_sequence(xr, lengths, batch_first=True))
    out, _ = pad_packed_sequence(out, batch_first=True, total_length=L)
    w = torch.randn(B, L, 2 * H, dtype=torch.float64)
    (out * w).sum().backward()

    dev = "cuda:0"
    mod = nn.LSTM(E, H, 1, batch_first=True, bidirectional=True).to(dev)
    mod.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    xh = x.float().to(dev).requires_grad_()
    oh = DF.bilstm(xh, lengths.to(dev), mod)
    (oh * w.float().to(dev)).sum().backward()

    def close(a, b, tol, what):
        err = float((a.detach().double().cpu() - b.detach()).abs().max())
        scale = max(float(b.abs().max()), 1e-3)
        assert err <= tol * scale, "%s: %.3e > %.1e*%.3g" % (what, err, tol, scale)

    close(oh, out, 2e-5, "out")
    close(xh.grad, xr.grad, 1e-4, "dx")
    for k, p in mod.named_parameters():
        close(p.grad, dict(ref.named_parameters())[k].grad, 1e-4, k)


def test_sentence_vector_rides_in_the_step_kernels():
    """qvec = [out[b][0] ; out[b][len_b-1]] written by the forward steps, and its gradient added to dout rows 0 / len_b-1 as
    the backward steps read them: bit-identical to the separate gather / scatter-add kernels (language_module.py:48-54)."""
    from drn_amd import functional as DF
    from drn_amd import ops
    torch.manual_seed(1)
    dev = "cuda:0"
    B, L, E, H = 6, 5, 20, 64
    lens = torch.tensor([5, 4, 3, 2, 1, 1], dtype=torch.int64, device=dev)
    mod = nn.LSTM(E, H, 1, batch_first=True, bidirectional=True).to(dev)
    params = DF._lstm_param_list(mod)
    emb_tm = torch.randn(L * B, E, device=dev)
    qvec = torch.full((B, 4 * H), float("nan"), device=dev)
    out, saved = DF._lstm_forward(emb_tm, lens, params, B, L, qvec=qvec)
    want = torch.empty_like(qvec)
    ops.qe_qvec_fwd(out, lens, want, B, L, 2 * H)
    assert torch.equal(qvec, want)
    dout = torch.randn(B, L, 2 * H, device=dev)
    dqvec = torch.randn(B, 4 * H, device=dev)
    leaves = []
    d_a, _ = DF._lstm_backward(dout, emb_tm, lens, params, saved, B, L, leaves, dqvec=dqvec)
    dgates_a = leaves[0]["dY"].clone()
    dout_b = dout.clone()
    ops.qe_qvec_bwd(dqvec, lens, dout_b, B, L, 2 * H)
    leaves_b = []
    d_b, _ = DF._lstm_backward(dout_b, emb_tm, lens, params, saved, B, L, leaves_b)
    assert torch.equal(d_a, d_b) and torch.equal(dgates_a, leaves_b[0]["dY"])


@pytest.mark.parametrize("B,L,lens", [(32, 8, None), (7, 5, [5, 5, 4, 3, 2, 1, 1]), (40, 3, None), (32, 24, None), (16, 48, None)])
def test_low_precision_recurrent_products_track_the_fp32_kernels(B, L, lens):
    """The bf16 model's BiLSTM (lowp): forward on v_mfma_f32_16x16x32_f16 with an fp16 copy of the hidden state (W_hh rounded to fp16
    in registers), backward on v_mfma_f32_16x16x32_bf16 with bf16 copies of W_hh^T and of the gate gradients; fp32 accumulation and
    fp32 states everywhere -- against the exact-fp32 kernels on the same inputs."""
    from drn_amd import functional as DF
    torch.manual_seed(3)
    dev = "cuda:0"
    E, H = 300, 512
    if lens is None:
        lens = sorted(torch.randint(1, L + 1, (B,)).tolist(), reverse=True)
        lens[0] = L
    lengths = torch.tensor(lens, dtype=torch.int64, device=dev)
    mod = nn.LSTM(E, H, 1, batch_first=True, bidirectional=True).to(dev)
    x = torch.randn(B, L, E, device=dev)
    w = torch.randn(B, L, 2 * H, device=dev)
    res = {}
    for lowp in (False, True):
        mod.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_()
        o = DF.bilstm(xi, lengths, mod, lowp=lowp)
        (o * w).sum().backward()
        res[lowp] = (o.detach().clone(), xi.grad.clone(), {k: p.grad.clone() for k, p in mod.named_parameters()})
    o32, dx32, g32 = res[False]
    o16, dx16, g16 = res[True]
    assert not torch.equal(o32, o16)                                  # the bf16 path really ran
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-12))
    # fp16 forward: unit roundoff 2^-11 on both operands.  The bound does not grow with the sequence (24 and 48 steps are three / six
    # times Charades-STA's longest query): the gates squash the state every step, rounding errors do not accumulate through the recurrence
    assert rel(o16, o32) <= 2e-3, rel(o16, o32)
    assert rel(dx16, dx32) <= 2e-2, rel(dx16, dx32)
    for k in g32:
        assert rel(g16[k], g32[k]) <= 2e-2, (k, rel(g16[k], g32[k]))
    for b in range(B):                                                # padded positions stay exact zeros
        assert float(o16[b, lens[b]:].abs().max() if lens[b] < L else 0.0) == 0.0


@pytest.mark.parametrize("lowp", [False, True])
def test_padded_time_steps_change_nothing(lowp):
    """The hipGraph trainer pads queries to a fixed length, the eager one does not: extra all-invalid time steps must leave every
    output BIT-identical (the cell backward is inlined into two kernels, and which one handles a step depends on the padding --
    round 3 found them one ulp apart until FMA contraction was switched off in that function)."""
    from drn_amd import functional as DF
    torch.manual_seed(0)
    dev = "cuda:0"
    B, E, H = 4, 300, 512
    lens = torch.tensor([3, 2, 2, 1], dtype=torch.int64, device=dev)
    mod = nn.LSTM(E, H, 1, batch_first=True, bidirectional=True).to(dev)
    params = DF._lstm_param_list(mod)
    x3 = torch.randn(3 * B, E, device=dev)
    g = torch.Generator().manual_seed(1)
    d3 = torch.randn(B, 3, 2 * H, generator=g).to(dev)
    dqvec = torch.randn(B, 4 * H, generator=g).to(dev)
    res = []
    for L in (3, 4, 8):
        emb = torch.zeros(L * B, E, device=dev)
        emb[:3 * B] = x3
        qvec = torch.empty(B, 4 * H, device=dev)
        out, saved = DF._lstm_forward(emb, lens, params, B, L, qvec=qvec)
        dout = torch.zeros(B, L, 2 * H, device=dev)
        dout[:, :3] = d3
        leaves = []
        demb, _ = DF._lstm_backward(dout, emb, lens, params, saved, B, L, leaves, dqvec=dqvec, lowp=lowp)
        res.append((out[:, :3].clone(), qvec.clone(), demb[:3 * B].clone(), leaves[0]["dY"][:3 * B].clone(), leaves[2]["dY"][:3 * B].clone()))
        assert float(out[:, 3:].abs().max() if L > 3 else 0.0) == 0.0 and float(demb[3 * B:].abs().max() if L > 3 else 0.0) == 0.0
    for r in res[1:]:
        for a, b in zip(res[0], r):
            assert torch.equal(a, b)


@pytest.mark.parametrize("B,L,lens", [(32, 8, None), (7, 5, [5, 5, 4, 3, 2, 1, 1]), (40, 3, None), (32, 4, None)])
def test_forward_in_one_launch_gives_the_bits_of_the_step_launches(monkeypatch, B, L, lens):
    """drn_lstm_seq_fwd: every step of the fp16-state forward in ONE launch (workgroups stay, hidden states handed over as fp16 words
    whose spare exponent bit carries the launch parity) -- against L drn_lstm_step_fwd launches: every output bit for bit, over several
    launches in a row (the parity alternates, stale data of the previous launch must never be taken) and with the sentence vector."""
    from drn_amd import functional as DF, ops
    torch.manual_seed(5)
    dev = "cuda:0"
    E, H = 300, 512
    if lens is None:
        lens = sorted(torch.randint(1, L + 1, (B,)).tolist(), reverse=True)
        lens[0] = L
    lengths = torch.tensor(lens, dtype=torch.int64, device=dev)
    mod = nn.LSTM(E, H, 1, batch_first=True, bidirectional=True).to(dev)
    params = DF._lstm_param_list(mod)

    def run(seq, emb):
        monkeypatch.setattr(ops, "LSTM_SEQ", seq)
        qvec = torch.full((B, 4 * H), float("nan"), device=dev)
        out, saved = DF._lstm_forward(emb, lengths, params, B, L, qvec=qvec, lowp=True)
        torch.cuda.synchronize()
        return [out.clone(), qvec.clone()] + [t.clone() for t in saved]
    for rep in range(5):                               # five launches on the same exchange buffer, different inputs each time
        emb = torch.randn(L * B, E, device=dev) * (1.0 + rep)
        a = run(True, emb)
        b = run(False, emb)
        for i, (x, y) in enumerate(zip(a, b)):
            if i == 2 or i == 3:                       # cseq / gates: slot 0 of cseq is never written by either path
                x, y = (x[:, 1:], y[:, 1:]) if i == 2 else (x, y)
            assert torch.equal(x, y), "rep %d tensor %d: max |d| = %g" % (rep, i, float((x - y).abs().max()))
    assert ops.lstm_seq_fwd_timeouts() == 0
    assert any(k[0][0] == "lstm_seq" for k in ops._persistent if isinstance(k[0], tuple)), "the one-launch path did not run"
