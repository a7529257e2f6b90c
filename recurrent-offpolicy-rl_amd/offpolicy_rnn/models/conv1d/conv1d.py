"""`conv1d[_K]` layer - depthwise causal convolution over time as a (finite-memory) recurrent layer, followed by the
position-wise feed-forward block (reference offpolicy_rnn/models/conv1d/conv1d.py:5-66).

hidden = the last K - 1 (masked) inputs, time-major [K - 1, C]; returned with the reference's `(B, 1, -1)` shape (:45),
which has the same memory layout as `(1, B, -1)`.  T > 1: the window is prepended and the whole row goes through the
causal-conv kernel; T == 1: one rolling-window kernel (`ops.conv_step`)."""
import torch
import torch.nn as nn

from ..gilr.gilr import PositionWiseFeedForward
from ...hip import ops


class Conv1d(nn.Module):
    def __init__(self, in_channels, out_channels, d_conv=4, bias=True, ff=True):
        super().__init__()
        assert in_channels == out_channels
        self.in_channels, self.out_channels, self.d_conv = in_channels, out_channels, d_conv
        self.conv1d = nn.Conv1d(in_channels, out_channels, kernel_size=d_conv, groups=in_channels, padding=0, bias=bias)
        self.desired_hidden_dim = in_channels * (d_conv - 1)
        self.use_ff = ff
        if ff:
            self.ff = PositionWiseFeedForward(out_channels, 0.0)

    def forward(self, x, hidden=None, mask=None):
        B, T, C = x.shape
        K = self.d_conv
        if hidden is None:
            hidden = torch.zeros((B, (K - 1) * C), device=x.device, dtype=x.dtype)
        hidden = hidden.reshape(B, (K - 1) * C)
        if mask is not None:
            x = x * mask
        if T == 1 and not torch.is_grad_enabled():
            y, tail = ops.conv_step(x[:, 0], hidden, self.conv1d.weight, self.conv1d.bias, K, 'kd', False)
            y = y.unsqueeze(1)
        else:
            rows = torch.cat((hidden.reshape(B, K - 1, C), x), dim=1)           # :33
            y = ops.causal_conv1d_fn(rows, self.conv1d.weight, self.conv1d.bias, None, False)[:, K - 1:]
            tail = rows[:, rows.shape[1] - (K - 1):].reshape(B, -1)             # :37
        if self.use_ff:
            y = self.ff(y)
        return y, tail.reshape(B, 1, -1)
