"""`gilr_lstm` layer - two stacked gated linear recurrences with LSTM-style gates
(reference offpolicy_rnn/models/gilr_lstm/gilr_lstm.py:13-75).

  stage 1: v = tanh(.), f = sigmoid(.) (1 - start):   c_t = f_t c_{t-1} + (1 - f_t) v_t        (as `gilr`)
  stage 2: [f, i, o, z] = middle_proj(c):              m_t = f_t m_{t-1} + (1 - f_t) (i_t z_t),  out = o_t m_t
Both recurrences run on the time-parallel scan kernel (`ops.gilr_scan`); hidden = (c | m)."""
import torch
import torch.nn as nn

from ..ensemble_linear_model import EnsembleLinear
from ...hip import ops
from ..linear import Linear


class GILRLSTMLayer(nn.Module):
    def __init__(self, input_dim, output_dim, factor=1, dropout=0.2, batch_first=True):
        super().__init__()
        assert batch_first
        self.d_model = output_dim
        self.in_proj = EnsembleLinear(input_dim, self.d_model * factor, 2, desire_ndim=4)
        self.middle_proj = EnsembleLinear(self.d_model * factor, self.d_model * factor, 4, desire_ndim=4)
        self.out_proj = Linear(self.d_model * factor, self.d_model * factor)
        self.layer_norm = nn.LayerNorm(factor * self.d_model)      # constructed (state_dict parity) but unused, as upstream
        self.swish = nn.SiLU()

    def rnn_parameters(self):
        return list(self.parameters(True))

    def forward(self, x, hidden=None, rnn_start=None):
        u = self.in_proj(x)                                         # [2, B, T, C]
        c0 = m0 = None
        if hidden is not None:
            c0, m0 = (t.contiguous() for t in hidden[0].chunk(2, dim=-1))
        c = ops.gilr_scan(u[0], u[1], rnn_start, c0, True)
        g = self.middle_proj(c)                                     # [4, B, T, C]
        f, i, o, z = torch.sigmoid(g[0]), torch.sigmoid(g[1]), torch.sigmoid(g[2]), torch.tanh(g[3])
        m = ops.gilr_scan(i * z, f, rnn_start, m0, False)
        out = self.out_proj(m * o)
        return out, torch.cat((c[:, -1:, :], m[:, -1:, :]), dim=-1).transpose(0, 1)
