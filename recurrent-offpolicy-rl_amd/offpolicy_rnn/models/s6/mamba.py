"""`mamba[_s<N>][_c<K>][_noff]` layer - the reference's first ("s6") Mamba variant: ONE residual block
[norm -> mixer -> + x] -> feed-forward, whose selective scan carries reset flags and an explicit hidden state
(reference offpolicy_rnn/models/s6/mamba.py:14-238; scan semantics selective_scan/cpu_scan.py:6-62).

The mixer is the same computation as `smamba`'s (in_proj -> masked causal conv + SiLU -> x_proj / dt_proj -> selective
scan with D skip, gated by silu(res) -> out_proj), so it runs on the same HIP kernels; what differs is the block wiring
(post-residual, RMSNorm eps 1e-5, FF with LayerNorm) and the hidden layout: (ssm state [Di, N] | conv tail [K - 1, Di]
time-major).  T > 1 rows start from the zero state - what the full-trajectory trainers pass (`make_init_state`), and how
the reference's own `smamba` treats multi-token calls (smamba/mamba.py:159-162); a carried state enters through T == 1
steps (rollouts), which use the one-token kernels."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..gilr.gilr import PositionWiseFeedForward
from ...hip import ops
from ..linear import Linear


class RMSNorm(nn.Module):
    def __init__(self, d_model: int, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(d_model))

    def forward(self, x):
        return ops.rms_norm_fn(x, self.weight, None, eps=self.eps)


class MambaBlock(nn.Module):
    def __init__(self, d_model, bias=False, dt_rank='auto', expand=2, d_state=16, d_conv=4):
        super().__init__()
        self.d_inner = int(expand * d_model)
        self.dt_rank = int(math.ceil(d_model / 16)) if dt_rank == 'auto' else dt_rank
        self.d_conv, self.d_state = d_conv, d_state
        assert d_conv >= 2, 'the conv-free variant (d_conv < 1) of the reference is not built'
        self.in_proj = Linear(d_model, self.d_inner * 2, bias=bias)
        self.conv1d = nn.Conv1d(self.d_inner, self.d_inner, kernel_size=d_conv, groups=self.d_inner, padding=0, bias=True)
        self.x_proj = Linear(self.d_inner, self.dt_rank + d_state * 2, bias=False)
        self.dt_proj = Linear(self.dt_rank, self.d_inner, bias=True)
        self._init_dt_proj_weight()
        self.ssm_hidden_dim = self.d_inner * d_state
        self.conv_hidden_dim = self.d_inner * (d_conv - 1)
        self.desired_hidden_dim = self.ssm_hidden_dim + self.conv_hidden_dim
        self.A_log = nn.Parameter(torch.log(torch.arange(1, d_state + 1, dtype=torch.float32)).repeat(self.d_inner, 1).contiguous())
        self.A_log._no_weight_decay = True
        self.D = nn.Parameter(torch.ones(self.d_inner))
        self.D._no_weight_decay = True
        self.out_proj = Linear(self.d_inner, d_model, bias=bias)

    def _init_dt_proj_weight(self, dt_scale=1.0, dt_max=0.1, dt_min=0.001, dt_init_floor=1e-4):
        with torch.no_grad():
            std = self.dt_rank ** -0.5 * dt_scale
            nn.init.uniform_(self.dt_proj.weight, -std, std)
            dt = torch.exp(torch.rand(self.d_inner) * (math.log(dt_max) - math.log(dt_min)) + math.log(dt_min)).clamp(min=dt_init_floor)
            self.dt_proj.bias.copy_(dt + torch.log(-torch.expm1(-dt)))
        self.dt_proj.bias._no_reinit = True

    def forward(self, x, hidden=None, rnn_start=None, mask=None, grad_detach=None):
        B, T, _ = x.shape
        Di, N, K, R = self.d_inner, self.d_state, self.d_conv, self.dt_rank
        xz = ops.linear(x, self.in_proj.weight, self.in_proj.bias)                   # [B, T, 2 Di]
        if T == 1 and hidden is not None and not torch.is_grad_enabled():
            h = hidden.reshape(B, -1)
            xi = xz[:, 0, :Di] if mask is None else xz[:, 0, :Di] * mask[:, 0]
            xc, tail = ops.conv_step(xi, h[:, Di * N:], self.conv1d.weight, self.conv1d.bias, K, 'kd', True)
            x_db = ops.linear(xc, self.x_proj.weight)
            y, state = ops.selective_state_update(h[:, :Di * N], xc, x_db, self.dt_proj.weight, self.dt_proj.bias, self.A_log,
                                                  self.D, xz[:, 0, Di:])
            out = ops.linear(y, self.out_proj.weight, self.out_proj.bias).unsqueeze(1)
            return out, torch.cat((state, tail), dim=-1).reshape(B, 1, -1)
        xi = xz[..., :Di] if mask is None else xz[..., :Di] * mask
        xc = ops.causal_conv1d_fn(xi, self.conv1d.weight, self.conv1d.bias, None, True)
        x_db = ops.linear(xc, self.x_proj.weight)
        dt = ops.linear(x_db[..., :R], self.dt_proj.weight)                            # bias enters as delta_bias
        A = -torch.exp(self.A_log.float())
        y, last = ops.selective_scan_tm(xc, dt, A, x_db[..., R:R + N], x_db[..., R + N:], self.D.float(), xz[..., Di:],
                                        self.dt_proj.bias.float(), rnn_start, True, True)
        out = ops.linear(y, self.out_proj.weight, self.out_proj.bias)
        tail = xi[:, T - (K - 1):] if T >= K - 1 else torch.cat((xi.new_zeros(B, K - 1 - T, Di), xi), dim=1)
        return out, torch.cat((last.reshape(B, 1, -1), tail.reshape(B, 1, -1)), dim=-1)


class MambaResidualBlock(nn.Module):
    def __init__(self, input_dim, output_dim, bias=False, dt_rank='auto', expand=2, d_state=16, d_conv=4, use_ff=True,
                 norm_type='rms'):
        super().__init__()
        assert input_dim == output_dim
        make = {'ln': lambda: nn.LayerNorm(output_dim), 'rms': lambda: RMSNorm(output_dim), 'none': nn.Identity}[norm_type]
        self.mixer = MambaBlock(input_dim, bias, dt_rank, expand, d_state, d_conv)
        self.norm = make()
        self.use_ff = use_ff
        if use_ff:
            self.ff = PositionWiseFeedForward(output_dim, 0.0)
        else:
            self.ff = Linear(output_dim, output_dim, bias=False)
            self.norm_f = make()

    def forward(self, x, hidden=None, rnn_start=None, mask=None, grad_detach=None):
        out, hidden = self.mixer(self.norm(x), hidden, rnn_start, mask, grad_detach)
        out = out + x
        return (self.ff(out) if self.use_ff else self.ff(self.norm_f(out))), hidden
