"""Query encoder (reference: model/language_module.py:9-98, model/ops.py:16-25,74-85) -- SURVEY row 8f-4.

The BiLSTM recurrence runs in drn_amd/csrc/lstm.hip with sequence lengths on the device (no packed sequences,
no host-side control flow), so the whole training step is hipGraph-capturable; the embedding lookup, the small
Linears, the masked softmax and the (B,1,L)x(B,L,1024) products are stock PyTorch-ROCm library calls."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as DF


class Linear(nn.Linear):
    """model/ops.py:16-25: TensorFlow-style xavier uniform, zero bias."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        bound = np.sqrt(3. / ((self.in_features + self.out_features) / 2.))
        nn.init.uniform_(self.weight, -bound, bound)
        if self.bias is not None:
            nn.init.constant_(self.bias, 0.)


class TextualAttention(nn.Module):
    """model/language_module.py:65-74: constructed by the reference but never called (keeps checkpoint keys)."""

    def __init__(self, hidden_dim=1024):
        super(TextualAttention, self).__init__()
        self.hidden_dim = hidden_dim
        self.q_dim = hidden_dim * 2
        self.W1 = nn.Linear(self.hidden_dim, 1)
        self.W2 = nn.Linear(self.q_dim, self.hidden_dim)
        self.W3 = nn.Linear(self.q_dim, self.q_dim)


class QueryEncoder(nn.Module):
    def __init__(self, vocab_size, hidden_dim=512, embed_dim=300, num_layers=1, bidirection=True):
        super(QueryEncoder, self).__init__()
        self.hidden_dim = hidden_dim
        self.embed_dim = embed_dim
        self.embedding = nn.Embedding(vocab_size + 1, embed_dim, padding_idx=0)
        self.biLSTM = nn.LSTM(embed_dim, self.hidden_dim, num_layers, dropout=0.0, batch_first=True,
                              bidirectional=bidirection)
        self.textualAttention = TextualAttention()
        self.qInput = Linear(self.hidden_dim * 4, self.hidden_dim)
        for t in range(3):
            setattr(self, "qInput%d" % t, Linear(self.hidden_dim, self.hidden_dim * 2))
        self.cmd_inter2logits = Linear(self.hidden_dim * 2, 1)

    def forward(self, query_tokens, query_length):
        emb = self.embedding(query_tokens)
        lengths = query_length if query_length.device == emb.device else query_length.to(emb.device)
        # (B, Lmax, 2H), zeros at padded positions like pad_packed_sequence(batch_first=True)
        output = DF.bilstm(emb, lengths, self.biLSTM)
        B, Lmax, H2 = output.shape
        lengths = lengths.to(torch.int64)
        last = output.gather(1, (lengths - 1).view(B, 1, 1).expand(B, 1, H2)).squeeze(1)
        q_vector = torch.cat((output[:, 0], last), dim=-1)                # language_module.py:48-54
        base = F.relu(self.qInput(q_vector))
        pad = torch.arange(Lmax, device=output.device).view(1, Lmax) >= lengths.view(B, 1)
        # language_module.py:27-36 for the three "commands" at once (same arithmetic, batched):
        #   raw_att[b,t,l] = sum_c (q_cmd[b,t,c] * w[c]) * output[b,l,c] + bias  ==  cmd_inter2logits(q_cmd[:,None,:] * output)
        W3 = torch.cat([getattr(self, "qInput%d" % t).weight for t in range(3)], dim=0)
        b3 = torch.cat([getattr(self, "qInput%d" % t).bias for t in range(3)], dim=0)
        q_cmd = F.linear(base, W3, b3).view(B, 3, H2)
        raw_att = torch.baddbmm(self.cmd_inter2logits.bias.view(1, 1, 1), q_cmd * self.cmd_inter2logits.weight.view(1, 1, H2),
                                output.transpose(1, 2))                  # (B, 3, Lmax)
        att = F.softmax(raw_att.masked_fill(pad[:, None, :], -1e30), dim=-1)
        cmds = torch.bmm(att, output)                                    # (B, 3, 2H)
        return [cmds[:, t] for t in range(3)]
