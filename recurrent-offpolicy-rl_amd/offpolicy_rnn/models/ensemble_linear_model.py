"""EnsembleLinear: E independent Linear layers evaluated as batched GEMMs (reference
offpolicy_rnn/models/ensemble_linear_model.py:8-69).  Parameter names / shapes (`weight` [E, in, out], `bias`
[E, 1, out]) and the shape polymorphism steered by `desire_ndim` follow the reference.

The contractions are plain library GEMMs (hipBLASLt through torch), arranged so that no operand is ever copied:
  * shared input (every member sees the same rows): ONE GEMM  [M, in] x [in, E*out]  with the bias fused (addmm); the result
    is handed on as an [E, M, out] *view* (strides (out, E*out, 1)) that the next layer's strided-batched GEMM consumes
    directly; the backward is again one GEMM each for dX (which also sums over the ensemble) and dW;
  * per-member input [E, M, in]: baddbmm with fused bias; dW = X^T G and dX = G W^T as strided-batched GEMMs on views.
torch's generic einsum/bmm autograd materialised a transposed [E, in, M] copy of the activations per layer and step
(547 MB at config 2) plus separate bias-add kernels; this custom Function removes both."""
import torch
import torch.nn as nn


class _SharedInput(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x2, weight, bias):
        E, n_in, n_out = weight.shape
        w_cat = weight.permute(1, 0, 2).reshape(n_in, E * n_out)                 # [in, E*out] (weights only: tiny copy)
        if bias is not None:
            y2 = torch.addmm(bias.reshape(E * n_out), x2, w_cat)
        else:
            y2 = torch.mm(x2, w_cat)
        ctx.save_for_backward(x2, w_cat)
        ctx.dims = (E, n_in, n_out, bias is not None)
        return y2.view(-1, E, n_out).transpose(0, 1)                              # [E, M, out] view

    @staticmethod
    def backward(ctx, g):
        x2, w_cat = ctx.saved_tensors
        E, n_in, n_out, has_bias = ctx.dims
        g2 = g.transpose(0, 1).reshape(-1, E * n_out)                             # a view when g kept y's strides
        dx = torch.mm(g2, w_cat.t()) if ctx.needs_input_grad[0] else None        # sums over the ensemble
        dw = torch.mm(x2.t(), g2).view(n_in, E, n_out).permute(1, 0, 2) if ctx.needs_input_grad[1] else None
        db = g2.sum(dim=0).view(E, 1, n_out) if has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db


class _PerMember(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x3, weight, bias):
        y = torch.baddbmm(bias, x3, weight) if bias is not None else torch.bmm(x3, weight)
        ctx.save_for_backward(x3, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, g):
        x3, weight = ctx.saved_tensors
        dx = torch.bmm(g, weight.transpose(1, 2)) if ctx.needs_input_grad[0] else None
        dw = torch.bmm(x3.transpose(1, 2), g) if ctx.needs_input_grad[1] else None
        db = g.sum(dim=1, keepdim=True) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db


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
        b = self.bias if self.use_bias else None
        if shared:
            lead = tuple(x.shape[:-1])
            y = _SharedInput.apply(x.reshape(-1, n_in), W, b)
        else:
            lead = tuple(x.shape[1:-1])
            y = _PerMember.apply(x.reshape(E, -1, n_in), W, b)
        return y.reshape((E,) + lead + (n_out,))                                   # a view: only the row axis is split
