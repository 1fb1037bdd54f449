"""Language-guided pooling (reference: model/LGP.py:3-51; instantiated nowhere in the reference, kept as a standalone op)."""
import torch
import torch.nn as nn

from .. import functional as DF


class LGP(nn.Module):
    compute_dtype = torch.float32

    def __init__(self, input_dim=1024, query_dim=1024, use_bn=True):
        super(LGP, self).__init__()
        # model/LGP.py:11-25: use_bn only decides whether the 1x1 conv carries a bias; the BatchNorm is appended either way
        conv = nn.Conv1d(query_dim, input_dim, kernel_size=1, stride=1, padding=0, dilation=1, bias=not use_bn)
        nn.init.kaiming_uniform_(conv.weight, a=1)
        self.query_fc = nn.Sequential(conv, nn.BatchNorm1d(input_dim))

    def forward(self, inputs, query):
        """inputs (B, C, t), query (B, Cq) -> (B, C, t/2)."""
        x = DF.as_nlc(inputs, self.compute_dtype)
        out = DF.lgp(x, query, self.query_fc[0], self.query_fc[1], self.training, self.compute_dtype)
        return out.permute(0, 2, 1)
