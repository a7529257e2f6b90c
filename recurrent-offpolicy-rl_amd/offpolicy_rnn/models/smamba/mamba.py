"""`smamba` layer - Mamba block list with start resets and validity mask (reference offpolicy_rnn/models/smamba/mamba.py).

Everything stays TOKEN-MAJOR [B, T', C]: `in_proj` is one GEMM whose output's two column halves are x and z, the
depthwise conv, `x_proj`, `dt_proj` and the selective scan all consume / produce column slices of row-major
matrices by stride.  The reference instead works channel-major (B, d_inner, L) and pays, per block and pass, the
`rearrange(...).contiguous()` copies of B and C, a materialised (B, d_inner, L) fp32 `start` tensor
(mamba.py:183) and a transposing copy of y in front of `out_proj` (mamba.py:251).

Parameter names and shapes equal the reference's (`layers.{i}.mixer.in_proj.weight`, `...conv1d.weight` [Di,1,K],
`...A_log`, `norm_f.weight`, `head.weight`, ...), so its checkpoints load unchanged."""
import math
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

from ...hip import ops
from ..linear import Linear


class RMSNorm(nn.Module):
    def __init__(self, hidden_size, eps=1e-5, device=None, dtype=None):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(hidden_size, device=device, dtype=dtype))
        self.register_parameter('bias', None)


class PositionWiseFeedForward(nn.Module):
    def __init__(self, d_model, dropout=0.0, eps=1e-5):
        super().__init__()
        self.w_1 = Linear(d_model, d_model)
        self.w_2 = Linear(d_model, d_model)
        self.activation = nn.GELU()
        self.dropout = nn.Dropout(dropout)
        self.layer_norm = nn.LayerNorm(d_model, eps=eps)

    def forward(self, x):
        y = self.dropout(self.activation(self.w_1(x)))
        return self.layer_norm(self.dropout(self.w_2(y)) + x)


class Mamba(nn.Module):
    def __init__(self, d_model, d_state=16, d_conv=4, expand=2, dt_rank='auto', dt_min=0.001, dt_max=0.1,
                 dt_init='random', dt_scale=1.0, dt_init_floor=1e-4, conv_bias=True, bias=False, layer_idx=None):
        super().__init__()
        self.d_model, self.d_state, self.d_conv, self.expand = d_model, d_state, d_conv, expand
        self.d_inner = int(expand * d_model)
        self.dt_rank = math.ceil(d_model / 16) if dt_rank == 'auto' else dt_rank
        self.layer_idx = layer_idx
        self.conv_hidden_dim = self.d_inner * d_conv
        self.ssm_hidden_dim = self.d_inner * d_state
        self.desired_hidden_dim = self.conv_hidden_dim + self.ssm_hidden_dim
        self.in_proj = Linear(d_model, self.d_inner * 2, bias=bias)
        self.conv1d = nn.Conv1d(self.d_inner, self.d_inner, kernel_size=d_conv, groups=self.d_inner, padding=d_conv - 1,
                                bias=conv_bias)
        self.x_proj = Linear(self.d_inner, self.dt_rank + 2 * d_state, bias=False)
        self.dt_proj = Linear(self.dt_rank, self.d_inner, bias=True)
        std = self.dt_rank ** -0.5 * dt_scale
        if dt_init == 'constant':
            nn.init.constant_(self.dt_proj.weight, std)
        else:
            nn.init.uniform_(self.dt_proj.weight, -std, std)
        # bias such that softplus(bias) is log-uniform in [dt_min, dt_max]
        dt = torch.exp(torch.rand(self.d_inner) * (math.log(dt_max) - math.log(dt_min)) + math.log(dt_min)).clamp(min=dt_init_floor)
        with torch.no_grad():
            self.dt_proj.bias.copy_(dt + torch.log(-torch.expm1(-dt)))
        self.dt_proj.bias._no_reinit = True
        self.A_log = nn.Parameter(torch.log(torch.arange(1, d_state + 1, dtype=torch.float32)).repeat(self.d_inner, 1).contiguous())
        self.A_log._no_weight_decay = True
        self.D = nn.Parameter(torch.ones(self.d_inner))
        self.D._no_weight_decay = True
        self.out_proj = Linear(self.d_inner, d_model, bias=bias)

    def forward(self, x, hidden=None, rnn_start=None, mask=None):
        """x [B, T, D].  T > 1: whole packed rows from a zero state (training).  T == 1: stateful rollout step."""
        if x.shape[-2] == 1:
            return self._step(x, hidden)
        if self.in_proj.bias is None and self.out_proj.bias is None:
            out = ops.mamba_inner_fn(x, self.in_proj.weight, self.conv1d.weight, self.conv1d.bias, self.x_proj.weight,
                                     self.dt_proj.weight, self.dt_proj.bias, self.A_log, self.D, self.out_proj.weight, mask, rnn_start)
            if hidden is None:
                hidden = torch.zeros((1, x.shape[0], self.desired_hidden_dim), device=x.device)
            return out, hidden
        Di, N, R = self.d_inner, self.d_state, self.dt_rank
        xz = ops.linear(x, self.in_proj.weight, self.in_proj.bias)                     # [B, T, 2 Di]
        xc = ops.causal_conv1d_fn(xz[..., :Di], self.conv1d.weight, self.conv1d.bias, mask, True)
        x_dbl = ops.linear(xc, self.x_proj.weight)                                     # [B, T, R + 2N]
        dt = ops.linear(x_dbl[..., :R], self.dt_proj.weight)                           # bias enters as delta_bias
        A = -torch.exp(self.A_log.float())
        y = ops.selective_scan_tm(xc, dt, A, x_dbl[..., R:R + N], x_dbl[..., R + N:], self.D.float(), xz[..., Di:],
                                  self.dt_proj.bias.float(), rnn_start, True)
        out = ops.linear(y, self.out_proj.weight, self.out_proj.bias)
        if hidden is None:
            hidden = torch.zeros((1, x.shape[0], self.desired_hidden_dim), device=x.device)
        return out, hidden

    def _step(self, x, hidden):
        """One-token rollout update (reference mamba.py:257-305): conv window rolled left, newest tap last; the window
        roll + conv + SiLU and the dt_proj + softplus + state update + gate are one kernel each (ops.mamba_step)."""
        B = x.shape[0]
        if hidden is None:
            hidden = torch.zeros((1, B, self.desired_hidden_dim), device=x.device)
        xz = ops.linear(x[:, 0], self.in_proj.weight, self.in_proj.bias)    # M = B rows against the whole weight: the rows form of resel_gemm_f32x
        y, new_hidden = ops.mamba_step(hidden[0], xz, self.conv1d.weight, self.conv1d.bias, self.x_proj.weight,
                                       self.dt_proj.weight, self.dt_proj.bias, self.A_log, self.D, self.d_conv, self.d_state)
        out = ops.linear(y, self.out_proj.weight, self.out_proj.bias).unsqueeze(1)
        return out, new_hidden.unsqueeze(0)


def _init_weights(module, n_layer, n_residuals_per_layer=1):
    """GPT-2 style residual scaling of out_proj / fc2 and zero Linear biases (reference mamba.py:323-352)."""
    if isinstance(module, nn.Linear) and module.bias is not None and not getattr(module.bias, '_no_reinit', False):
        nn.init.zeros_(module.bias)
    for name, p in module.named_parameters():
        if name in ('out_proj.weight', 'fc2.weight'):
            nn.init.kaiming_uniform_(p, a=math.sqrt(5))
            with torch.no_grad():
                p /= math.sqrt(n_residuals_per_layer * n_layer)


class Block(nn.Module):
    """Add -> Norm -> Mixer with the running residual kept in fp32 (fused add+norm kernel)."""

    def __init__(self, dim, mixer_cls, norm_cls=nn.LayerNorm, fused_add_norm=True, residual_in_fp32=True):
        super().__init__()
        self.residual_in_fp32, self.fused_add_norm = residual_in_fp32, fused_add_norm
        self.mixer = mixer_cls(dim)
        self.norm = norm_cls(dim)

    def forward(self, hidden_states, residual=None, hidden=None, rnn_start=None, mask=None):
        fn = ops.rms_norm_fn if isinstance(self.norm, RMSNorm) else ops.layer_norm_fn
        hidden_states, residual = fn(hidden_states, self.norm.weight, self.norm.bias, residual=residual, prenorm=True,
                                     residual_in_fp32=self.residual_in_fp32, eps=self.norm.eps)
        out, hidden = self.mixer(hidden_states, hidden, rnn_start, mask)
        return out, hidden, residual


class BlockList(nn.Module):
    def __init__(self, block_num, dim, d_conv=4, d_state=16, fused_add_norm=True, rms_norm=True, residual_in_fp32=True,
                 use_ff=False):
        super().__init__()
        self.block_num, self.fused_add_norm, self.rms_norm = block_num, fused_add_norm, rms_norm
        self.norm_epsilon = 1e-8
        self.d_conv, self.residual_in_fp32, self.use_ff = d_conv, residual_in_fp32, use_ff
        norm_cls = partial(RMSNorm if rms_norm else nn.LayerNorm, eps=self.norm_epsilon)
        self.layers = nn.ModuleList([
            Block(dim, partial(Mamba, layer_idx=i, d_conv=d_conv, d_state=d_state), norm_cls=norm_cls,
                  fused_add_norm=fused_add_norm, residual_in_fp32=residual_in_fp32) for i in range(block_num)])
        self.desired_hidden_dim = self.layers[0].mixer.desired_hidden_dim * block_num
        if use_ff:
            self.head = PositionWiseFeedForward(d_model=dim, dropout=0.0, eps=self.norm_epsilon)
        else:
            self.head = Linear(dim, dim, bias=False)
            self.norm_f = norm_cls(dim)
        self.apply(partial(_init_weights, n_layer=block_num))

    def forward(self, x, hidden=None, rnn_start=None, mask=None, out_act=None):
        """out_act = 'elu': the caller's plain ELU behind this layer is applied by the head GEMM's epilogue (training passes, rnn_base.py)."""
        if hidden is None:
            hidden = torch.zeros((1, x.shape[0], self.desired_hidden_dim), device=x.device)
        states = torch.chunk(hidden, self.block_num, dim=-1)
        residual, outs = None, []
        for i, block in enumerate(self.layers):
            x, h, residual = block(x, residual, states[i], rnn_start, mask)
            outs.append(h)
        if not self.use_ff:
            fn = ops.rms_norm_fn if self.rms_norm else ops.layer_norm_fn
            x = fn(x, self.norm_f.weight, self.norm_f.bias, eps=self.norm_f.eps, residual=residual, prenorm=False,
                   residual_in_fp32=self.residual_in_fp32)
        else:
            x = x + residual
        # the bias-free head Linear through ops.linear (hand-written GEMM forward / input gradient / weight gradient on training passes)
        if out_act is not None:
            assert isinstance(self.head, nn.Linear) and x.shape[-2] > 1
            x = ops.linear_act(x, self.head.weight, None, out_act)
        else:
            x = ops.linear(x, self.head.weight, None) if isinstance(self.head, nn.Linear) and x.shape[-2] > 1 else self.head(x)
        # whole-row passes leave the state placeholders untouched: hand the caller's tensor back instead of re-assembling it (12.6 MB at config 2)
        return x, (hidden if all(o is st for o, st in zip(outs, states)) else torch.cat(outs, dim=-1))
