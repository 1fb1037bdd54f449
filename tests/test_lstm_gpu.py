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
