"""`gilr` layer - gated input linear RNN (reference offpolicy_rnn/models/gilr/gilr.py:13-81).

h_t = f_t h_{t-1} + (1 - f_t) v_t with v = tanh(.), f = sigmoid(.) (1 - rnn_start): the activations, the reset
folding and the recurrence run in ONE time-parallel HIP kernel (`ops.gilr_scan`); the reference needs three
element-wise passes plus a sequential Triton scan for the same thing."""
import torch
import torch.nn as nn

from ..ensemble_linear_model import EnsembleLinear
from ...hip import ops
from ..linear import Linear


class PositionWiseFeedForward(nn.Module):
    def __init__(self, d_model, dropout=0.1, eps=1e-5):
        super().__init__()
        self.w_1 = Linear(d_model, d_model)
        self.w_2 = Linear(d_model, d_model)
        self.activation = nn.GELU()
        self.dropout = nn.Dropout(dropout)
        self.layer_norm = nn.LayerNorm(d_model, eps=eps)

    def forward(self, x, out_act=None):
        """out_act='elu': the plain ELU behind the layer rides in the closing add + LayerNorm kernel (RNNBase, training passes)."""
        h1 = ops.linear(x, self.w_1.weight, self.w_1.bias)
        y = self.activation(h1)
        if y.is_cuda:
            ops.tag_amax(y, ops.amax_of(h1))          # |gelu(v)| = |v| Phi(v) <= |v|: the magnitude the w_1 product published bounds its GELU too, so
        y = self.dropout(y)                           # w_2 (and its weight gradient) run product mode 2 instead of 6 (66 -> 46 us at configs[4])
        return ops.layer_norm_fn(self.dropout(ops.linear(y, self.w_2.weight, self.w_2.bias)), self.layer_norm.weight, self.layer_norm.bias, residual=x,
                                 eps=self.layer_norm.eps, act=out_act)        # fused add + LayerNorm (+ ELU)


class GILRLayer(nn.Module):
    def __init__(self, input_dim, output_dim, factor=1, dropout=0.0, use_ff=True, batch_first=True):
        super().__init__()
        assert batch_first
        self.d_model = output_dim
        self.in_proj = EnsembleLinear(input_dim, self.d_model * factor, 2, desire_ndim=4)
        self.out_proj = Linear(self.d_model * factor, self.d_model * factor)
        self.dropout = nn.Dropout(dropout)
        self.layer_norm = nn.LayerNorm(factor * self.d_model)      # constructed (state_dict parity) but unused, as upstream
        self.use_ff = use_ff
        if use_ff:
            self.ff = PositionWiseFeedForward(self.d_model, dropout)

    def rnn_parameters(self):
        return list(self.parameters(True))

    def forward(self, x, hidden=None, rnn_start=None, out_act=None):
        u = self.in_proj(x)                                         # [2, B, T, C]
        h0 = None if hidden is None else hidden[0]
        h = ops.gilr_scan_members(u, rnn_start, h0, True)           # the two members read in place, one gradient tensor back
        assert out_act is None or self.use_ff
        out = ops.linear(h, self.out_proj.weight, self.out_proj.bias)
        if self.use_ff:
            out = self.ff(out, out_act)
        return out, h[:, -1:, :].transpose(0, 1)
