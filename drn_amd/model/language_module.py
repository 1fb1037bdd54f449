"""Query encoder (reference: model/language_module.py:9-98, model/ops.py:16-25,74-85) -- SURVEY row 8f-4.

The BiLSTM recurrence (drn_amd/csrc/lstm.hip) and the glue around it (drn_amd/csrc/qenc.hip: embedding gather /
scatter, sentence vector, the three attention commands) run with sequence lengths on the device -- no packed
sequences, no host-side control flow, so the whole training step is hipGraph-capturable; the handful of dense
products (input projection, qInput*, their gradients) run on the grouped exact-f32 MFMA kernels of drn_amd/csrc/qdense.hip
(`skinny_group_kernel`, `outer_wgrad_kernel`) over stacked weights -- no rocBLAS / hipBLASLt call anywhere in the step."""
import numpy as np
import torch
import torch.nn as nn

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
    # torch.bfloat16 (set by mainModel.set_compute_dtype), the numerical contract of the bf16 model's query side:
    #   forward recurrence: the hidden state is kept as an fp16 copy (|h| < 1, unit roundoff 2^-11) and multiplied on
    #     v_mfma_f32_16x16x32_f16 with W_hh rounded to fp16 in registers, fp32 accumulation, fp32 cell state / outputs;
    #   backward recurrence: bf16 MFMA with the optimizer-maintained bf16 copy of W_hh^T and bf16-rounded gate gradients (fp32
    #     accumulation); the weight-gradient products round their staged operands to bf16;
    #   every other product of the query side (input projection, qInput*, attention) stays exact fp32 in the forward pass.
    compute_dtype = torch.float32

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
        """language_module.py:38-63 -> the three attention commands [(B, 2H)] * 3, as one fused autograd node
        (drn_amd.functional._QueryEncoderFn: embedding, BiLSTM, sentence vector, qInput*, attention)."""
        return list(DF.query_encoder(query_tokens, query_length, self, lowp=self.compute_dtype == torch.bfloat16))
