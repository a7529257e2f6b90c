"""`gru` layer: torch.nn.GRU's parameters and formula, recurrence on the HIP step kernels.

The reference builds `torch.nn.GRU(in, out, batch_first=True)` (rnn_base.py:59,247) and calls it without reset or
mask handling (:453-454).  This module keeps the parameter names of nn.GRU (`weight_ih_l0`, `weight_hh_l0`,
`bias_ih_l0`, `bias_hh_l0`) so checkpoints load unchanged; the input projection is hoisted out of the time loop as
one GEMM over all B*T' tokens, the recurrence runs in `ops.gru_seq`."""
import torch
import torch.nn as nn

from ..hip import ops


class GRU(nn.Module):
    def __init__(self, input_size: int, hidden_size: int, batch_first: bool = True):
        super().__init__()
        assert batch_first
        self.input_size, self.hidden_size = input_size, hidden_size
        k = 1.0 / hidden_size ** 0.5
        self.weight_ih_l0 = nn.Parameter(torch.empty(3 * hidden_size, input_size).uniform_(-k, k))
        self.weight_hh_l0 = nn.Parameter(torch.empty(3 * hidden_size, hidden_size).uniform_(-k, k))
        self.bias_ih_l0 = nn.Parameter(torch.empty(3 * hidden_size).uniform_(-k, k))
        self.bias_hh_l0 = nn.Parameter(torch.empty(3 * hidden_size).uniform_(-k, k))

    def forward(self, x, hidden=None):
        """x [B, T, in]; hidden [1, B, H] -> (y [B, T, H], last hidden [1, B, H])."""
        gi = ops.linear(x, self.weight_ih_l0, self.bias_ih_l0)
        h0 = None if hidden is None else hidden[0]
        y = ops.gru_seq(gi, self.weight_hh_l0, self.bias_hh_l0, h0)
        return y, y[:, -1:, :].transpose(0, 1)
