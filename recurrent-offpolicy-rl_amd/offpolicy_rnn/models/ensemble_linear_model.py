"""EnsembleLinear: E independent Linear layers evaluated as one batched GEMM (reference
offpolicy_rnn/models/ensemble_linear_model.py:8-69).  Parameter names / shapes (`weight` [E, in, out], `bias`
[E, 1, out]) and the shape polymorphism steered by `desire_ndim` follow the reference; the contraction itself is a
plain library GEMM (`torch.matmul` -> hipBLASLt), with the (rows, T') axes flattened so that every call is one
strided-batched GEMM of E problems."""
import torch
import torch.nn as nn


class EnsembleLinear(nn.Module):
    def __init__(self, input_dim: int, output_dim: int, num_ensemble: int, bias: bool = True, desire_ndim: int = None):
        super().__init__()
        self.use_bias = bias
        self.desire_ndim = desire_ndim
        self.num_ensemble = num_ensemble
        self.weight = nn.Parameter(torch.zeros(num_ensemble, input_dim, output_dim))
        if bias:
            self.bias = nn.Parameter(torch.zeros(num_ensemble, 1, output_dim))
        nn.init.trunc_normal_(self.weight, std=1 / (2 * input_dim ** 0.5))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        W = self.weight
        E, n_in, n_out = W.shape
        nd = x.dim()
        shared = True                       # True: every member sees the same input (x has no ensemble axis)
        if nd == 3:
            shared = not ((self.desire_ndim is None or self.desire_ndim == 3) and x.shape[0] == E)
        elif nd == 4:
            shared = not ((self.desire_ndim is None or self.desire_ndim == 4) and x.shape[0] == E)
        elif nd == 5:
            shared = False
        if shared:
            lead = x.shape[:-1]
            y = torch.matmul(x.reshape(1, -1, n_in), W)                      # [E, rows, out]
            y = y.reshape((E,) + tuple(lead) + (n_out,))
        else:
            lead = x.shape[1:-1]
            y = torch.bmm(x.reshape(E, -1, n_in), W).reshape((E,) + tuple(lead) + (n_out,))
        if self.use_bias:
            y = y + self.bias.reshape((E,) + (1,) * (y.dim() - 2) + (n_out,))
        return y
