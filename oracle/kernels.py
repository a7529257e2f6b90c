"""Op-level CPU restatements: one function per C-ABI entry point of `include/resel_hip.h`.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Everything here is plain PyTorch on CPU tensors,
differentiable through autograd, written for clarity, not speed.  Layout convention = the product's
HBM layout: activations are TOKEN-MAJOR `[B, L, C]` (channel stride 1); per-token flags are `[B, L]`.

Citations are `path:line` inside the reference checkout (FanmingL/Recurrent-Offpolicy-RL @ 2024_10_08).
"""
import math
from typing import Optional, Tuple

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------
# smamba: selective scan  (reference spec: offpolicy_rnn/models/smamba/mamba_ssm/ops/
# selective_scan_interface_new.py:96-166 `selective_scan_ref`; reset semantics :133-135;
# the external `selective_scan_cuda` it stands in for is called at :47 / :72)
# --------------------------------------------------------------------------------------------
def selective_scan_ref(u, delta, A, Bm, Cm, D=None, z=None, delta_bias=None, start=None,
                       delta_softplus=True, h0=None):
    """u, delta, z: [B, L, Di]; A: [Di, N]; Bm, Cm: [B, L, N]; D, delta_bias: [Di]; start: [B, L] (1 = reset).

    h_t = exp(delta_t A) * (1 - start_t) * h_{t-1} + delta_t * Bm_t * u_t ;  y_t = <Cm_t, h_t> + D u_t ;
    out_t = y_t * silu(z_t).   Returns (out [B, L, Di], last_state [B, Di, N]).
    """
    Bsz, L, Di = u.shape
    N = A.shape[1]
    u = u.float()
    delta = delta.float()
    if delta_bias is not None:
        delta = delta + delta_bias.float()                        # :115-116
    if delta_softplus:
        delta = F.softplus(delta)                                 # :117-118
    h = torch.zeros(Bsz, Di, N, dtype=torch.float32) if h0 is None else h0.float()
    ys = []
    for t in range(L):
        dA = torch.exp(delta[:, t, :, None] * A[None].float())    # :132
        if start is not None:
            dA = dA * (1.0 - start[:, t].float())[:, None, None]  # :133-135
        dBu = (delta[:, t] * u[:, t])[:, :, None] * Bm[:, t].float()[:, None, :]   # :140
        h = dA * h + dBu                                          # :148
        ys.append((h * Cm[:, t].float()[:, None, :]).sum(-1))     # :153
    y = torch.stack(ys, dim=1)
    if D is not None:
        y = y + u * D.float()                                     # :162
    if z is not None:
        y = y * F.silu(z.float())                                 # :163-164
    return y, h


# --------------------------------------------------------------------------------------------
# smamba: depthwise causal conv1d + bias + SiLU on the masked input
# (reference: offpolicy_rnn/models/smamba/mamba.py:75-83 Conv1d(groups=d_inner, padding=d_conv-1),
#  :210-212 `x = mask * x; x = act(conv1d(x)[..., :seqlen])`; step form :264-271)
# --------------------------------------------------------------------------------------------
def causal_conv1d_silu_ref(x, w, bias=None, mask=None, activation=True):
    """x: [B, L, Di]; w: [Di, K]; bias: [Di]; mask: [B, L].  y_t = silu(b + sum_k w[:,k] * xm_{t-(K-1)+k})."""
    Bsz, L, Di = x.shape
    K = w.shape[1]
    xm = x if mask is None else x * mask[:, :, None]
    xp = F.pad(xm, (0, 0, K - 1, 0))                              # zero left pad in time
    y = torch.zeros_like(x)
    for k in range(K):
        y = y + xp[:, k:k + L, :] * w[:, k]
    if bias is not None:
        y = y + bias
    return F.silu(y) if activation else y


# --------------------------------------------------------------------------------------------
# smamba: fused residual-add + LayerNorm / RMSNorm
# (reference CPU spec: offpolicy_rnn/models/smamba/mamba_ssm/ops/triton/layernorm_cpu.py:6-19 (LN),
#  :22-35 (RMS); GPU Triton kernels layernorm.py:65,196 are out-of-tree-equivalent)
# --------------------------------------------------------------------------------------------
def add_layernorm_ref(x, residual, weight, bias, eps, rms=False):
    """x, residual: [..., C].  Returns (y, residual_out) with residual_out = x + residual (fp32)."""
    res = x.float() if residual is None else x.float() + residual.float()
    if rms:
        rstd = 1.0 / torch.sqrt(res.square().mean(-1, keepdim=True) + eps)      # layernorm_cpu.py:32
        y = res * rstd * weight
        if bias is not None:
            y = y + bias
    else:
        y = F.layer_norm(res, res.shape[-1:], weight, bias, eps)               # layernorm_cpu.py:16
    return y, res


# --------------------------------------------------------------------------------------------
# gilr: gated real linear recurrence
# (reference: offpolicy_rnn/models/gilr/gilr.py:44-67 activations + reset folding,
#  CPU scan offpolicy_rnn/models/gilr/scan_triton/real_rnn_tie_input_gate_cpu.py:4-14,
#  Triton kernel real_rnn_tie_input_gate.py:9-34 `h = (h - v) * f + v`)
# --------------------------------------------------------------------------------------------
def linrec_real_ref(v, f, start=None, h0=None, fuse_act=True):
    """v, f: [B, L, C] (pre-activation if fuse_act); start: [B, L]; h0: [B, C].

    fuse_act: v <- tanh(v), f <- sigmoid(f) * (1 - start)   (gilr.py:52-56)
    h_t = f_t * h_{t-1} + (1 - f_t) * v_t.   Returns (h_all [B, L, C], h_last [B, C]).
    """
    if fuse_act:
        v = torch.tanh(v)
        f = torch.sigmoid(f)
    if start is not None:
        f = f * (1.0 - start[:, :, None])
    Bsz, L, C = v.shape
    h = torch.zeros(Bsz, C, dtype=v.dtype) if h0 is None else h0
    out = []
    for t in range(L):
        h = h * f[:, t] + v[:, t] * (1.0 - f[:, t])               # real_rnn_tie_input_gate_cpu.py:11
        out.append(h)
    return torch.stack(out, dim=1), h


# --------------------------------------------------------------------------------------------
# lru: complex diagonal linear recurrence
# (reference: offpolicy_rnn/models/lru/lru.py:95-115 lambda/gamma + reset folding,
#  CPU scan offpolicy_rnn/models/lru/scan_triton/complex_rnn_cpu.py:4-26,
#  Triton kernel complex_rnn.py:44-87)
# --------------------------------------------------------------------------------------------
def linrec_complex_ref(vr, vi, lam_re, lam_im, start=None, h0r=None, h0i=None, gamma=None):
    """vr, vi: [B, L, C]; lam_re, lam_im, gamma: [C]; start: [B, L]; h0r/h0i: [B, C].

    v <- gamma * v (lru.py:103-105); f_t = lambda * (1 - start_t) (lru.py:112-115);
    h_t = f_t * h_{t-1} + v_t (complex).  Returns (hr, hi [B, L, C]).
    """
    if gamma is not None:
        vr = vr * gamma
        vi = vi * gamma
    Bsz, L, C = vr.shape
    hr = torch.zeros(Bsz, C, dtype=vr.dtype) if h0r is None else h0r
    hi = torch.zeros(Bsz, C, dtype=vr.dtype) if h0i is None else h0i
    outr, outi = [], []
    for t in range(L):
        keep = 1.0 if start is None else (1.0 - start[:, t])[:, None]
        fr = lam_re * keep
        fi = lam_im * keep
        nr = hr * fr - hi * fi + vr[:, t]                         # complex_rnn_cpu.py:17
        ni = hr * fi + hi * fr + vi[:, t]                         # complex_rnn_cpu.py:18
        hr, hi = nr, ni
        outr.append(hr)
        outi.append(hi)
    return torch.stack(outr, dim=1), torch.stack(outi, dim=1)


# --------------------------------------------------------------------------------------------
# gru: 1-layer GRU recurrence on a hoisted input projection
# (reference: torch.nn.GRU(batch_first=True) built at offpolicy_rnn/models/rnn_base.py:59,247 and
#  called without reset/mask handling at :453-454; gate order r,z,n, PyTorch definition)
# --------------------------------------------------------------------------------------------
def gru_seq_ref(gi, w_hh, b_hh, h0=None):
    """gi = x @ W_ih^T + b_ih: [B, L, 3H] (r,z,n blocks); w_hh: [3H, H]; b_hh: [3H]; h0: [B, H].

    r = sig(gi_r + W_hr h + b_hr); z = sig(gi_z + W_hz h + b_hz); n = tanh(gi_n + r*(W_hn h + b_hn));
    h' = (1 - z) * n + z * h.   Returns h_all [B, L, H].
    """
    Bsz, L, H3 = gi.shape
    H = H3 // 3
    h = torch.zeros(Bsz, H, dtype=gi.dtype) if h0 is None else h0
    out = []
    for t in range(L):
        gh = h @ w_hh.t() + b_hh
        r = torch.sigmoid(gi[:, t, :H] + gh[:, :H])
        zg = torch.sigmoid(gi[:, t, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, t, 2 * H:] + r * gh[:, 2 * H:])
        h = (1.0 - zg) * n + zg * h
        out.append(h)
    return torch.stack(out, dim=1)


# --------------------------------------------------------------------------------------------
# cgpt: packed var-len causal attention with ALiBi     *** PARITY UNPINNED ***
# (reference call sites only: offpolicy_rnn/models/flash_attention/TransformerFlashAttention.py:67-70
#  MHA(causal=True, use_alibi=True), packing :107-112, bf16 autocast :80-82.  `flash_attn` itself is
#  an unpinned, un-vendored dependency (requirement.txt:7); this restates its published semantics:
#  scores = q k^T / sqrt(d) - slope_h * (i - j) for j <= i, softmax in fp32, slopes 2^(-8 h / H),
#  h = 1..H for power-of-two H (flash_attn.ops... get_alibi_slopes).)
# --------------------------------------------------------------------------------------------
def alibi_slopes(nheads: int) -> torch.Tensor:
    def pow2(n):
        start = 2.0 ** (-(2.0 ** -(math.log2(n) - 3)))
        return [start * (start ** i) for i in range(n)]
    if math.log2(nheads).is_integer():
        s = pow2(nheads)
    else:
        c = 2 ** math.floor(math.log2(nheads))
        s = pow2(c) + pow2(2 * c)[0::2][: nheads - c]
    return torch.tensor(s, dtype=torch.float32)


# counter-keyed dropout masks.  The reference takes its masks from flash-attn's Philox stream (attention probabilities,
# MHA(dropout=p), TransformerFlashAttention.py:67-70) and from ATen's (nn.Dropout, :48,72) - neither stream is
# reproducible outside those libraries, so the product defines its own counter function (include/resel_hip.h,
# csrc/attention.hip, csrc/dropout.hip) and this is its restatement; what IS the reference's: keep probability 1 - p,
# kept values scaled by 1 / (1 - p), softmax statistics taken before the mask, flash-attn's 8-bit keep threshold.
_M32 = 0xFFFFFFFF
DROP_CQ, DROP_CK, DROP_CH = 0x9E3779B1, 0x85EBCA77, 0xC2B2AE3D


def _mix32(x):
    """`lowbias32` on uint64 numpy arrays holding 32-bit values (or python ints)."""
    x = x ^ (x >> 16); x = (x * 0x7FEB352D) & _M32
    x = x ^ (x >> 15); x = (x * 0x846CA68B) & _M32
    return x ^ (x >> 16)


def _stream_key(seed: int, offset: int, head_term: int) -> int:
    x = _mix32((head_term ^ (offset >> 32)) & _M32)
    x = _mix32(x ^ (offset & _M32))
    x = _mix32(x ^ ((seed >> 32) & _M32))
    return _mix32(x ^ (seed & _M32))


def attn_dropout_keep(seed: int, offset: int, H: int, q_tok0: int, n: int, p_drop: float):
    """bool [H, n, n] keep mask of one packed sequence whose first token has packed index q_tok0:
    keep(h, i, j) = byte (j & 3) of mix32((q_tok0 + i) * CQ ^ (j >> 2) * CK ^ head_key(h)) < floor((1 - p) * 255) + 1."""
    import numpy as np
    thr = int(math.floor((1.0 - p_drop) * 255.0)) + 1
    i = (np.arange(n, dtype=np.uint64) + np.uint64(q_tok0))[:, None]
    j = np.arange(n, dtype=np.uint64)[None, :]
    base = ((i * np.uint64(DROP_CQ)) & np.uint64(_M32)) ^ (((j >> np.uint64(2)) * np.uint64(DROP_CK)) & np.uint64(_M32))
    keep = np.empty((H, n, n), dtype=bool)
    for h in range(H):
        w = _mix32(base ^ np.uint64(_stream_key(seed, offset, (h * DROP_CH) & _M32)))
        keep[h] = ((w >> ((j & np.uint64(3)) * np.uint64(8))) & np.uint64(0xFF)) < np.uint64(thr)
    return torch.from_numpy(keep)


def dropout_keep(seed: int, offset: int, n: int, p_drop: float):
    """bool [n] keep mask of `resel_dropout`: 16-bit half (i & 1) of mix32((i >> 1) * CQ ^ key) < round((1 - p) * 65536)."""
    import numpy as np
    thr = int(round((1.0 - p_drop) * 65536.0))
    i = np.arange(n, dtype=np.uint64)
    w = _mix32((((i >> np.uint64(1)) * np.uint64(DROP_CQ)) & np.uint64(_M32)) ^ np.uint64(_stream_key(seed, offset, DROP_CH)))
    return torch.from_numpy(((w >> ((i & np.uint64(1)) * np.uint64(16))) & np.uint64(0xFFFF)) < np.uint64(thr))


def dropout_ref(x, p_drop: float, seed: int, offset: int):
    if p_drop <= 0.0:
        return x
    keep = dropout_keep(seed, offset, x.numel(), p_drop).view(x.shape)
    return torch.where(keep, x / (1.0 - p_drop), torch.zeros_like(x))


class DropCounter:
    """Mirror of the product's `ops.dropout_counter`: every draw returns (seed, offset) and advances the offset by 4."""
    def __init__(self, seed: int, offset: int = 0):
        self.seed, self.offset = int(seed), int(offset)

    def next(self):
        out = (self.seed, self.offset)
        self.offset += 4
        return out


def attention_alibi_varlen_ref(q, k, v, cu_seqlens, slopes=None, scale=None, p_drop=0.0, seed=0, offset=0, p_bf16=False):
    """q, k, v: [T, H, d] packed tokens; cu_seqlens: int [S+1].  Returns out [T, H, d] (fp32 math).
    p_drop > 0: the probabilities entering P V are masked by `attn_dropout_keep` and scaled by 1 / (1 - p).
    p_bf16: the rounding point every bf16 flash-attention forward has (upstream flash-attn semantics, third party): the
    un-normalised probabilities exp(s - rowmax) enter the P V product rounded to bf16 while the row sum that normalises the
    result is taken over the unrounded fp32 values."""
    T, H, d = q.shape
    scale = (1.0 / math.sqrt(d)) if scale is None else scale
    out = torch.zeros(T, H, d, dtype=torch.float32)
    cu = [int(c) for c in cu_seqlens]
    for s in range(len(cu) - 1):
        a, b = cu[s], cu[s + 1]
        n = b - a
        if n <= 0:
            continue
        qs, ks, vs = q[a:b].float(), k[a:b].float(), v[a:b].float()
        sc = torch.einsum('ihd,jhd->hij', qs, ks) * scale
        i = torch.arange(n)[:, None]
        j = torch.arange(n)[None, :]
        if slopes is not None:
            sc = sc - slopes.float()[:, None, None] * (i - j).abs().float()[None]
        sc = sc.masked_fill((j > i)[None], float('-inf'))
        if p_bf16:
            e = torch.exp(sc - sc.max(dim=-1, keepdim=True).values)
            inv = 1.0 / e.sum(dim=-1, keepdim=True)
            e = e.to(torch.bfloat16).float()
            if p_drop > 0.0:
                e = torch.where(attn_dropout_keep(seed, offset, H, a, n, p_drop), e, torch.zeros_like(e))
                inv = inv / (1.0 - p_drop)
            out[a:b] = torch.einsum('hij,jhd->ihd', e, vs) * inv.permute(1, 0, 2)
            continue
        p = torch.softmax(sc, dim=-1)
        if p_drop > 0.0:
            p = torch.where(attn_dropout_keep(seed, offset, H, a, n, p_drop), p / (1.0 - p_drop), torch.zeros_like(p))
        out[a:b] = torch.einsum('hij,jhd->ihd', p, vs)
    return out


# --------------------------------------------------------------------------------------------
# one-token rollout step  (reference: offpolicy_rnn/models/smamba/mamba.py:257-305 CPU branch ==
# mamba_ssm/ops/triton/selective_state_update.py:123-154 `selective_state_update_ref`;
# flash_attn MHA with inference_params: append k, v to the cache, attend over positions 0..pos)
# --------------------------------------------------------------------------------------------
def mamba_step_ref(conv_state, ssm_state, xz, conv_w, conv_b, xproj_w, dt_w, dt_b, A_log, D):
    """conv_state [B, Di, K], ssm_state [B, Di, N], xz [B, 2 Di] -> (y [B, Di], conv_state', ssm_state')."""
    Di = xz.shape[-1] // 2
    x, z = xz[:, :Di], xz[:, Di:]                                              # :261
    conv_state = torch.roll(conv_state, shifts=-1, dims=-1)                    # :265
    conv_state = torch.cat((conv_state[:, :, :-1], x.unsqueeze(-1)), dim=-1)   # :266
    x = torch.sum(conv_state * conv_w, dim=-1)                                 # :268
    if conv_b is not None:
        x = x + conv_b
    x = F.silu(x)                                                              # :271
    x_db = F.linear(x, xproj_w)                                                # :281
    R, N = dt_w.shape[1], ssm_state.shape[-1]
    dt, Bm, Cm = x_db[:, :R], x_db[:, R:R + N], x_db[:, R + N:]
    dt = F.softplus(F.linear(dt, dt_w) + dt_b)                                 # :284,290
    A = -torch.exp(A_log.float())                                              # :285
    dA = torch.exp(dt.unsqueeze(-1) * A)                                       # :291
    dB = dt.unsqueeze(-1) * Bm.unsqueeze(1)                                    # :292
    ssm_state = ssm_state * dA + x.unsqueeze(-1) * dB                          # :293
    y = (ssm_state * Cm.unsqueeze(1)).sum(-1) + D * x                          # :294-295
    return y * F.silu(z), conv_state, ssm_state                                # :296


def attn_decode_ref(q, k_cache, v_cache, pos, slopes=None, scale=None):
    """q [B, H, d] at position `pos`; k_cache, v_cache [B, S, H, d] already holding positions 0..pos.  -> [B, H, d]."""
    d = q.shape[-1]
    scale = (1.0 / math.sqrt(d)) if scale is None else scale
    k, v = k_cache[:, :pos + 1].float(), v_cache[:, :pos + 1].float()
    sc = torch.einsum('bhd,bjhd->bhj', q.float(), k) * scale
    if slopes is not None:
        j = torch.arange(pos + 1, dtype=torch.float32)
        sc = sc - slopes.float()[None, :, None] * (pos - j)[None, None, :]
    return torch.einsum('bhj,bjhd->bhd', torch.softmax(sc, dim=-1), v)


# --------------------------------------------------------------------------------------------
# ensemble (batched) linear  (reference: offpolicy_rnn/models/ensemble_linear_model.py:29-60)
# --------------------------------------------------------------------------------------------
def ensemble_linear_ref(x, weight, bias=None, desire_ndim=None):
    """weight: [E, in, out]; bias: [E, 1, out].  Shape polymorphism follows ensemble_linear_model.py:36-49."""
    E = weight.shape[0]
    if x.dim() == 2:
        y = torch.einsum('ij,bjk->bik', x, weight)
    elif x.dim() == 3:
        if (desire_ndim is None or desire_ndim == 3) and x.shape[0] == E:
            y = torch.einsum('bij,bjk->bik', x, weight)
        else:
            y = torch.einsum('cij,bjk->bcik', x, weight)
    elif x.dim() == 4:
        if (desire_ndim is None or desire_ndim == 4) and x.shape[0] == E:
            y = torch.einsum('cbij,cjk->cbik', x, weight)
        else:
            y = torch.einsum('cdij,bjk->bcdik', x, weight)
    else:
        y = torch.einsum('bcdij,bjk->bcdik', x, weight)
    if bias is not None:
        b = bias
        if y.dim() == 4:
            b = b.unsqueeze(1)
        elif y.dim() == 5:
            b = b.unsqueeze(1).unsqueeze(1)
        y = y + b
    return y


# --------------------------------------------------------------------------------------------
# SAC / TD3 head + target + loss arithmetic
# --------------------------------------------------------------------------------------------
LOG_STD_MIN, LOG_STD_MAX = -20.0, 2.0      # contextual_sac_policy_single_head.py:12-13


def tanh_gaussian_ref(mean, logstd, noise):
    """offpolicy_rnn/policy_value_models/contextual_sac_policy_single_head.py:109-123.

    Returns (tanh(mean), tanh(mean + noise*std), log_prob [.., 1])."""
    logstd = torch.clamp(logstd, LOG_STD_MIN, LOG_STD_MAX)
    pre = mean + noise * logstd.exp()
    logp = (-0.5 * noise.pow(2) - (logstd + 0.5 * math.log(2 * math.pi))).sum(-1, keepdim=True)
    logp = logp - (2 * (-pre - F.softplus(-2 * pre) + math.log(2))).sum(-1, keepdim=True)
    return torch.tanh(mean), torch.tanh(pre), logp


def sac_target_ref(next_q_subset, next_logp, reward, done, alpha, gamma, qmin, qmax):
    """offpolicy_rnn/algorithm/sac_full_length_rnn_redq.py:28-33 (TD3: td3_full_length_rnn_redq.py:29-35
    with next_logp=None).  next_q_subset: [m, R, L, 1] (already restricted to the REDQ subset)."""
    mn = next_q_subset.min(dim=0).values
    if next_logp is not None:
        mn = mn - alpha * next_logp
    if qmin is not None or qmax is not None:
        mn = mn.clamp(min=qmin, max=qmax)                          # utility/q_value_guard.py:22-27
    return reward + (1.0 - done) * gamma * mn


def q_loss_ref(q, target, mask, valid_num):
    """offpolicy_rnn/algorithm/sac_full_length_rnn_ensembleQ.py:105-114 + :80-81.  q: [E, R, L, 1]."""
    return ((q - target.unsqueeze(0)).pow(2).sum(0) * mask).sum() / valid_num


def soft_update_ref(target, online, tau):
    """offpolicy_rnn/models/rnn_base.py:490-491:  target <- tau * target + (1 - tau) * online."""
    return target * tau + (1.0 - tau) * online


def adamw_ref(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, wd=0.0):
    """torch.optim.AdamW single-tensor update (the reference's optimizer, algorithm/sac.py:61)."""
    p = p * (1.0 - lr * wd)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)) + eps
    p = p - (lr / bc1) * m / denom
    return p, m, v
