"""`lru` layer - complex diagonal Linear Recurrent Unit (reference offpolicy_rnn/models/lru/lru.py:17-188).

h_t = lambda (1 - rnn_start_t) h_{t-1} + gamma (a_t + i b_t).  The reference materialises lambda per token
([B, T', C] x 2) to fold the resets in and scans sequentially; `ops.complex_scan` takes lambda / gamma per channel,
folds resets and the gamma scaling inside the kernel and scans time-parallel."""
import math

import torch
import torch.nn as nn

from ..ensemble_linear_model import EnsembleLinear
from ..gilr.gilr import PositionWiseFeedForward
from ...hip import ops


class LRULayer(nn.Module):
    def __init__(self, input_dim, output_dim, dropout=0.0, batch_first=True, use_ff=True, squash_inproj=False):
        super().__init__()
        assert batch_first, 'LRU only support batch_first==True'
        self.d_model = output_dim
        self.in_proj = EnsembleLinear(input_dim, self.d_model, num_ensemble=3, desire_ndim=4, bias=True)
        self.middle_proj = EnsembleLinear(self.d_model, self.d_model, num_ensemble=2, desire_ndim=4, bias=True)
        self.dropout = nn.Dropout(dropout)
        self.params_log = nn.Parameter(self._init_params(self.d_model))
        self.use_ff, self.squash_inproj = use_ff, squash_inproj
        if use_ff:
            self.ff = PositionWiseFeedForward(self.d_model, dropout)

    @staticmethod
    def _init_params(c, r_min=0.9, r_max=0.999):
        """|lambda| uniform on the ring [r_min, r_max], phase uniform, gamma normalising (arXiv 2303.06349 sec. 3.2.2)."""
        u1, u2 = torch.rand(c), torch.rand(c)
        nu_log = torch.log(-0.5 * torch.log(u1 * (r_max ** 2 - r_min ** 2) + r_min ** 2))
        theta_log = torch.log(u2 * math.pi * 2)
        gamma_log = torch.log(torch.sqrt(1 - torch.exp(-torch.exp(nu_log)) ** 2))
        return torch.vstack((nu_log, theta_log, gamma_log))

    def rnn_parameters(self):
        return self.parameters(recurse=True)

    def forward(self, x, hidden=None, rnn_start=None, grad_detach=None, out_act=None):
        u = self.in_proj(x)                                         # [3, B, T, C]
        if self.squash_inproj:
            u = torch.tanh(u)
        h0r = h0i = None
        if hidden is not None:
            h0r, h0i = hidden[0].chunk(2, dim=-1)
        # lambda = exp(-exp(nu_log)) e^{i exp(theta_log)}, gamma = exp(gamma_log): one launch (`ops.lru_params`); the members of u are read in
        # place, (Re h | Im h) comes out stacked, u[2] is handed through: one gradient tensor in u's layout comes back
        h2, u2 = ops.complex_scan_members(u, ops.lru_params(self.params_log), rnn_start, h0r, h0i)
        hr, hi = h2[0], h2[1]
        out = ops.SubAddMembers.apply(self.middle_proj(h2), u2)
        assert out_act is None or self.use_ff
        if self.use_ff:
            out = self.ff(out, out_act)
        hidden = torch.cat((hr[:, -1:, :], hi[:, -1:, :]), dim=-1).transpose(0, 1)
        return out, hidden
