"""Functional CPU restatement of the reference's network stack (TEST INFRASTRUCTURE ONLY).

Parameters live in plain dicts keyed exactly like the reference's `state_dict()`s, so a reference
checkpoint / golden fixture drops in unchanged:

    ContextualModel.state_dict()  ->  {module_name: {param_key: tensor}}      (contextual_model.py:165-169)

Layers covered: fc, efc-<E>, gru, gilr, lru, smamba_*, cgpt_*  (the layer ids of BASELINE.json).
Citations are `path:line` in the reference checkout.
"""
import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from . import kernels as K


# ------------------------------------------------------------------------------------------------
# layer-id grammar (offpolicy_rnn/models/rnn_base.py:101-247)
# ------------------------------------------------------------------------------------------------
def parse_layer_id(layer_id: str) -> dict:
    if layer_id == 'fc':
        return dict(kind='fc')
    if layer_id.startswith('efc'):
        return dict(kind='efc', ensemble=int(layer_id.split('-')[-1]))
    if layer_id in ('gru', 'gilr', 'lru', 'gilr_lstm'):
        return dict(kind=layer_id)
    if layer_id.startswith('conv1d'):                               # rnn_base.py:227-234
        return dict(kind='conv1d', d_conv=int(layer_id.split('_')[-1]) if '_' in layer_id else 4)
    if layer_id.startswith('mamba'):                                # rnn_base.py:118-135
        cfg = dict(kind='mamba', d_conv=4, d_state=16, use_ff=True)
        for c in layer_id.split('_')[1:]:
            if c.startswith('s'):
                cfg['d_state'] = int(c[1:])
            elif c.startswith('c'):
                cfg['d_conv'] = int(c[1:])
            elif c.startswith('no'):
                if c[2:] == 'ff':
                    cfg['use_ff'] = False
            else:
                raise ValueError(f'Pattern {c} has not been implemented!')
        return cfg
    if layer_id.startswith('smamba'):                               # rnn_base.py:137-163
        cfg = dict(kind='smamba', d_conv=4, d_state=16, block_num=2, rms_norm=True, use_ff=False)
        for c in layer_id.split('_')[1:]:
            if c.startswith('s'):
                cfg['d_state'] = int(c[1:])
            elif c.startswith('c'):
                cfg['d_conv'] = int(c[1:])
            elif c.startswith('b'):
                cfg['block_num'] = int(c[1:])
            elif c.startswith('n'):
                cfg['rms_norm'] = False if c[1:] == 'ln' else True
            elif c.startswith('f'):
                if c[1:] == 'f':
                    cfg['use_ff'] = True
            else:
                raise ValueError(f'Pattern {c} has not been implemented!')
        return cfg
    if layer_id.startswith('cgpt'):                                 # rnn_base.py:186-210
        cfg = dict(kind='cgpt', nhead=8, nlayer=4, pdrop=0.1, maxlength=1024, ln=True)
        for c in layer_id.split('_')[1:]:
            if c.startswith('h'):
                cfg['nhead'] = int(c[1:])
            elif c.startswith('l'):
                cfg['nlayer'] = int(c[1:])
            elif c.startswith('p'):
                cfg['pdrop'] = float(c[1:])
            elif c.startswith('ml'):
                cfg['maxlength'] = int(c[2:])
            elif c.startswith('rms'):
                cfg['ln'] = False
            else:
                raise ValueError(f'Pattern {c} has not been implemented!')
        return cfg
    raise NotImplementedError(layer_id)


def is_rnn(layer_id: str) -> bool:
    return parse_layer_id(layer_id)['kind'] not in ('fc', 'efc')


def hidden_size_of(layer_id: str, in_dim: int, out_dim: int) -> int:
    """rnn_hidden_state_input_size (rnn_base.py:107-247)."""
    c = parse_layer_id(layer_id)
    if c['kind'] in ('lru', 'gilr_lstm'):
        return out_dim * 2
    if c['kind'] == 'conv1d':
        return in_dim * (c['d_conv'] - 1)                           # conv1d/conv1d.py:21
    if c['kind'] == 'mamba':
        return in_dim * 2 * c['d_state'] + in_dim * 2 * (c['d_conv'] - 1)     # s6/mamba.py:105
    if c['kind'] == 'smamba':
        return (in_dim * 2 * c['d_conv'] + in_dim * 2 * c['d_state']) * c['block_num']   # smamba/mamba.py:70-72,446
    if c['kind'] == 'cgpt':
        return c['maxlength']
    return out_dim


ACT = {'tanh': torch.tanh, 'relu': F.relu, 'sigmoid': torch.sigmoid, 'leaky_relu': F.leaky_relu,
       'linear': lambda x: x, 'elu': F.elu, 'gelu': F.gelu}          # rnn_base.py:45-53


class Flags:
    """Side-channel of RNNHidden (offpolicy_rnn/models/RNNHidden.py:36-62)."""
    def __init__(self, rnn_start=None, mask=None, seqlens=None, dropout=None):
        self.rnn_start = rnn_start      # [B, L, 1]
        self.mask = mask                # [B, L, 1]
        self.seqlens = seqlens          # int [B, L] per-row sequence-length table (cgpt)
        self.dropout = dropout          # K.DropCounter in a training-mode pass of a p > 0 cgpt layer, else None


# ------------------------------------------------------------------------------------------------
# sequence layers
# ------------------------------------------------------------------------------------------------
def _ff_block(p, pre, x, eps=1e-5):
    """PositionWiseFeedForward: Linear-GELU-Linear + residual LayerNorm (gilr.py:70-81, lru.py:176-188,
    smamba/mamba.py:528-539)."""
    x_ = F.gelu(F.linear(x, p[pre + 'w_1.weight'], p[pre + 'w_1.bias']))
    y = F.linear(x_, p[pre + 'w_2.weight'], p[pre + 'w_2.bias']) + x
    return F.layer_norm(y, y.shape[-1:], p[pre + 'layer_norm.weight'], p[pre + 'layer_norm.bias'], eps)


def gru_layer(p, pre, x, flags=None):
    """torch.nn.GRU(batch_first=True), h0 = 0, no reset handling (rnn_base.py:453-454)."""
    gi = F.linear(x, p[pre + 'weight_ih_l0'], p[pre + 'bias_ih_l0'])
    return K.gru_seq_ref(gi, p[pre + 'weight_hh_l0'], p[pre + 'bias_hh_l0'])


def gru_layer_aten(p, pre, x, flags=None):
    """Same layer through ATen's fused CPU GRU - used for the timed CPU baseline."""
    B = x.shape[0]
    H = p[pre + 'weight_hh_l0'].shape[1]
    h0 = torch.zeros(1, B, H, dtype=x.dtype)
    w = [p[pre + 'weight_ih_l0'], p[pre + 'weight_hh_l0'], p[pre + 'bias_ih_l0'], p[pre + 'bias_hh_l0']]
    y, _ = torch._VF.gru(x, h0, w, True, 1, 0.0, False, False, True)
    return y


def gilr_layer(p, pre, x, flags=None):
    """offpolicy_rnn/models/gilr/gilr.py:44-67."""
    u = K.ensemble_linear_ref(x, p[pre + 'in_proj.weight'], p[pre + 'in_proj.bias'], desire_ndim=4)
    start = None if flags is None or flags.rnn_start is None else flags.rnn_start[..., 0]
    h, _ = K.linrec_real_ref(u[0], u[1], start, fuse_act=True)
    out = F.linear(h, p[pre + 'out_proj.weight'], p[pre + 'out_proj.bias'])
    return _ff_block(p, pre + 'ff.', out)


def lru_layer(p, pre, x, flags=None):
    """offpolicy_rnn/models/lru/lru.py:70-174."""
    u = K.ensemble_linear_ref(x, p[pre + 'in_proj.weight'], p[pre + 'in_proj.bias'], desire_ndim=4)
    params = torch.exp(p[pre + 'params_log'])                       # lru.py:95
    nu, theta, gamma = params[0], params[1], params[2]
    mag = torch.exp(-nu)                                            # lambda = exp(-nu + i theta), lru.py:99
    lam_re, lam_im = mag * torch.cos(theta), mag * torch.sin(theta)
    start = None if flags is None or flags.rnn_start is None else flags.rnn_start[..., 0]
    hr, hi = K.linrec_complex_ref(u[0], u[1], lam_re, lam_im, start, gamma=gamma)
    out = torch.stack((hr, hi), dim=0)
    out = K.ensemble_linear_ref(out, p[pre + 'middle_proj.weight'], p[pre + 'middle_proj.bias'], desire_ndim=4)
    out = out[0] - out[1] + u[2]                                    # lru.py:167
    return _ff_block(p, pre + 'ff.', out)


def gilr_lstm_layer(p, pre, x, flags=None, h0=None):
    """offpolicy_rnn/models/gilr_lstm/gilr_lstm.py:39-75.  h0 [B, 2C] = (stage-1 state | stage-2 state).  -> (y, hT)."""
    u = K.ensemble_linear_ref(x, p[pre + 'in_proj.weight'], p[pre + 'in_proj.bias'], desire_ndim=4)
    start = None if flags is None or flags.rnn_start is None else flags.rnn_start[..., 0]
    C = u.shape[-1]
    c0, m0 = (None, None) if h0 is None else (h0[:, :C], h0[:, C:])
    c, c_last = K.linrec_real_ref(u[0], u[1], start, c0, fuse_act=True)                      # :48-58
    g = K.ensemble_linear_ref(c, p[pre + 'middle_proj.weight'], p[pre + 'middle_proj.bias'], desire_ndim=4)
    f, i, o, z = torch.sigmoid(g[0]), torch.sigmoid(g[1]), torch.sigmoid(g[2]), torch.tanh(g[3])    # :59-62
    m, m_last = K.linrec_real_ref(i * z, f, start, m0, fuse_act=False)                       # :63-70
    y = F.linear(m * o, p[pre + 'out_proj.weight'], p[pre + 'out_proj.bias'])                 # :71-72
    return y, torch.cat((c_last, m_last), dim=-1)


def conv1d_layer(p, pre, x, flags=None, h0=None):
    """offpolicy_rnn/models/conv1d/conv1d.py:27-50: depthwise conv over (hidden ++ masked x), no activation, then FF.
    h0 [B, (K-1) C] time-major.  -> (y, hT)."""
    B, L, C = x.shape
    w = p[pre + 'conv1d.weight']                                    # [C, 1, K]
    Kw = w.shape[-1]
    hid = torch.zeros(B, Kw - 1, C) if h0 is None else h0.reshape(B, Kw - 1, C)
    if flags is not None and flags.mask is not None:
        x = x * flags.mask                                          # :30-31
    rows = torch.cat((hid, x), dim=1)                               # :32
    y = F.conv1d(rows.transpose(1, 2), w, p.get(pre + 'conv1d.bias'), groups=C)[:, :, :L].transpose(1, 2)   # :33-35
    return _ff_block(p, pre + 'ff.', y), rows[:, rows.shape[1] - (Kw - 1):].reshape(B, -1)


def s6_mamba_layer(p, pre, x, cfg, flags=None, h0=None):
    """MambaResidualBlock.forward over MambaBlock.forward / .ssm (offpolicy_rnn/models/s6/mamba.py:41-67,142-237) with
    the sequential scan of selective_scan/cpu_scan.py:6-62.  h0 [B, Di N + (K-1) Di] = (ssm | conv tail).  -> (y, hT)."""
    B, L, D = x.shape
    N, Kw = cfg['d_state'], cfg['d_conv']
    mp = pre + 'mixer.'
    h = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5) * p[pre + 'norm.weight']     # RMSNorm :240-250
    xz = F.linear(h, p[mp + 'in_proj.weight'])
    Di = xz.shape[-1] // 2
    xi, res = xz[..., :Di], xz[..., Di:]
    ssm0 = torch.zeros(B, Di, N) if h0 is None else h0[:, :Di * N].reshape(B, Di, N)
    conv0 = torch.zeros(B, Kw - 1, Di) if h0 is None else h0[:, Di * N:].reshape(B, Kw - 1, Di)
    if flags is not None and flags.mask is not None:
        xi = xi * flags.mask                                        # :131-132
    rows = torch.cat((conv0, xi), dim=1)
    xc = F.conv1d(rows.transpose(1, 2), p[mp + 'conv1d.weight'], p[mp + 'conv1d.bias'], groups=Di)[:, :, :L].transpose(1, 2)
    conv_tail = rows[:, rows.shape[1] - (Kw - 1):]
    xc = F.silu(xc)                                                 # :178
    x_db = F.linear(xc, p[mp + 'x_proj.weight'])
    R = x_db.shape[-1] - 2 * N
    delta = F.softplus(F.linear(x_db[..., :R], p[mp + 'dt_proj.weight'], p[mp + 'dt_proj.bias']))   # :229
    A = -torch.exp(p[mp + 'A_log'].float())
    Bm, Cm = x_db[..., R:R + N], x_db[..., R + N:]
    start = torch.zeros(B, L) if flags is None or flags.rnn_start is None else flags.rnn_start[..., 0]
    state, ys = ssm0, []
    for t in range(L):                                              # cpu_scan.py:41-57
        dA = torch.exp(delta[:, t, :, None] * A) * (1 - start[:, t])[:, None, None]
        state = dA * state + (delta[:, t] * xc[:, t])[:, :, None] * Bm[:, t, None, :]
        ys.append((state * Cm[:, t, None, :]).sum(-1))
    y = torch.stack(ys, dim=1) + xc * p[mp + 'D']
    y = y * F.silu(res)                                             # :181
    out = F.linear(y, p[mp + 'out_proj.weight']) + x                # :183, :62
    if cfg['use_ff']:
        out = _ff_block(p, pre + 'ff.', out)
    else:
        out = out * torch.rsqrt(out.pow(2).mean(-1, keepdim=True) + 1e-5) * p[pre + 'norm_f.weight']
        out = F.linear(out, p[pre + 'ff.weight'])
    return out, torch.cat((state.reshape(B, -1), conv_tail.reshape(B, -1)), dim=-1)


def mamba_mixer(p, pre, x, cfg, start, mask):
    """Mamba.forward_sequential, d_conv > 4 branch (offpolicy_rnn/models/smamba/mamba.py:166-255)."""
    N = cfg['d_state']
    xz = F.linear(x, p[pre + 'in_proj.weight'])                     # :175-179 (no bias)
    Di = xz.shape[-1] // 2
    xc, z = xz[..., :Di], xz[..., Di:]                              # :208
    w = p[pre + 'conv1d.weight'][:, 0, :]
    xc = K.causal_conv1d_silu_ref(xc, w, p[pre + 'conv1d.bias'], mask)          # :210-212
    x_dbl = F.linear(xc, p[pre + 'x_proj.weight'])                  # :231
    R = x_dbl.shape[-1] - 2 * N
    dt = F.linear(x_dbl[..., :R], p[pre + 'dt_proj.weight'])        # :233 (bias goes in as delta_bias)
    Bm, Cm = x_dbl[..., R:R + N], x_dbl[..., R + N:]
    A = -torch.exp(p[pre + 'A_log'].float())                        # :187
    y, _ = K.selective_scan_ref(xc, dt, A, Bm, Cm, p[pre + 'D'].float(), z,
                                p[pre + 'dt_proj.bias'].float(), start, True)    # :238-250
    return F.linear(y, p[pre + 'out_proj.weight'])                  # :252


def smamba_layer(p, pre, x, cfg, flags=None, semantics='gpu'):
    """BlockList.forward (offpolicy_rnn/models/smamba/mamba.py:492-526) over Block.forward (:382-412).

    semantics='gpu'      : forward_sequential with start resets + conv input mask (the training path on GPU)
    semantics='cpu_step' : what the reference does on CPU tensors - the per-step loop (:134-147) that
                           IGNORES rnn_start and mask."""
    start = mask = None
    if semantics == 'gpu' and flags is not None:
        start = None if flags.rnn_start is None else flags.rnn_start[..., 0]
        mask = None if flags.mask is None else flags.mask[..., 0]
    eps = 1e-8                                                      # :425
    rms = cfg['rms_norm']
    residual = None
    h = x
    for i in range(cfg['block_num']):
        bp = f'{pre}layers.{i}.'
        h, residual = K.add_layernorm_ref(h, residual, p[bp + 'norm.weight'], p.get(bp + 'norm.bias'), eps, rms)
        h = mamba_mixer(p, bp + 'mixer.', h, cfg, start, mask)
    if not cfg['use_ff']:
        h, _ = K.add_layernorm_ref(h, residual, p[pre + 'norm_f.weight'], p.get(pre + 'norm_f.bias'), eps, rms)
        return F.linear(h, p[pre + 'head.weight'])                  # :451,524
    h = h + residual
    return _ff_block(p, pre + 'head.', h, eps)


def smamba_layer_step(p, pre, x, hidden, cfg):
    """One rollout token through BlockList (mamba.py:492-526) with Mamba.forward's T == 1 branch (:134-157):
    x [B, D], hidden [B, blocks * Di * (K + N)] -> (y [B, D], new hidden)."""
    eps, rms = 1e-8, cfg['rms_norm']
    K_, N = cfg['d_conv'], cfg['d_state']
    residual, h, outs = None, x, []
    chunks = torch.chunk(hidden, cfg['block_num'], dim=-1)                     # :500
    for i in range(cfg['block_num']):
        bp = f'{pre}layers.{i}.'
        h, residual = K.add_layernorm_ref(h, residual, p[bp + 'norm.weight'], p.get(bp + 'norm.bias'), eps, rms)
        mp = bp + 'mixer.'
        xz = F.linear(h, p[mp + 'in_proj.weight'])
        B, Di = xz.shape[0], xz.shape[1] // 2
        conv = chunks[i][:, :Di * K_].reshape(B, Di, K_)                       # :138-141
        ssm = chunks[i][:, Di * K_:].reshape(B, Di, N)                         # :142-143
        y, conv, ssm = K.mamba_step_ref(conv, ssm, xz, p[mp + 'conv1d.weight'][:, 0, :], p[mp + 'conv1d.bias'],
                                        p[mp + 'x_proj.weight'], p[mp + 'dt_proj.weight'], p[mp + 'dt_proj.bias'],
                                        p[mp + 'A_log'], p[mp + 'D'])
        h = F.linear(y, p[mp + 'out_proj.weight'])
        outs.append(torch.cat((conv.reshape(B, -1), ssm.reshape(B, -1)), dim=-1))   # :153-156
    if not cfg['use_ff']:
        h, _ = K.add_layernorm_ref(h, residual, p[pre + 'norm_f.weight'], p.get(pre + 'norm_f.bias'), eps, rms)
        h = F.linear(h, p[pre + 'head.weight'])
    else:
        h = _ff_block(p, pre + 'head.', h + residual, eps)
    return h, torch.cat(outs, dim=-1)


def rollout_layer(p, lid, x, h0, pre='layer_list.0.'):
    """What T successive one-token `meta_forward` calls with a carried hidden state compute for a single-layer RNNBase
    (rnn_base.py:437-457; the rollout of algorithm/sac.py:319-326): x [B, T, D], h0 [B, hidden] -> (y [B, T, D], hT).
    gru / gilr / lru are pure recurrences, so the T steps are one pass from h0; smamba goes through Mamba.step."""
    c = parse_layer_id(lid)
    if c['kind'] == 'gru':
        gi = F.linear(x, p[pre + 'weight_ih_l0'], p[pre + 'bias_ih_l0'])
        y = K.gru_seq_ref(gi, p[pre + 'weight_hh_l0'], p[pre + 'bias_hh_l0'], h0)
        return y, y[:, -1]
    if c['kind'] == 'gilr':
        u = K.ensemble_linear_ref(x, p[pre + 'in_proj.weight'], p[pre + 'in_proj.bias'], desire_ndim=4)
        h, last = K.linrec_real_ref(u[0], u[1], None, h0, fuse_act=True)                     # gilr.py:48-62
        return _ff_block(p, pre + 'ff.', F.linear(h, p[pre + 'out_proj.weight'], p[pre + 'out_proj.bias'])), last
    if c['kind'] == 'lru':
        u = K.ensemble_linear_ref(x, p[pre + 'in_proj.weight'], p[pre + 'in_proj.bias'], desire_ndim=4)
        params = torch.exp(p[pre + 'params_log'])
        nu, theta, gamma = params[0], params[1], params[2]
        mag = torch.exp(-nu)
        C = h0.shape[-1] // 2                                                                  # hidden = (real | imag), lru.py:122-124
        hr, hi = K.linrec_complex_ref(u[0], u[1], mag * torch.cos(theta), mag * torch.sin(theta), None,
                                      h0[:, :C], h0[:, C:], gamma=gamma)
        out = K.ensemble_linear_ref(torch.stack((hr, hi), dim=0), p[pre + 'middle_proj.weight'], p[pre + 'middle_proj.bias'],
                                    desire_ndim=4)
        return _ff_block(p, pre + 'ff.', out[0] - out[1] + u[2]), torch.cat((hr[:, -1], hi[:, -1]), dim=-1)
    if c['kind'] == 'gilr_lstm':
        return gilr_lstm_layer(p, pre, x, None, h0)
    if c['kind'] == 'conv1d':
        return conv1d_layer(p, pre, x, None, h0)
    if c['kind'] == 'mamba':
        return s6_mamba_layer(p, pre, x, c, None, h0)
    if c['kind'] == 'smamba':
        ys, h = [], h0
        for t in range(x.shape[1]):
            y, h = smamba_layer_step(p, pre, x[:, t], h, c)
            ys.append(y)
        return torch.stack(ys, dim=1), h
    raise NotImplementedError(lid)


def seqlens_to_cu(seqlens: torch.Tensor):
    """Per-row length table -> (token indices, cu_seqlens) like flash_attn.bert_padding.
    unpad_input_for_concatenated_sequences (called at TransformerFlashAttention.py:107)."""
    B, L = seqlens.shape
    idx, cu = [], [0]
    for b in range(B):
        pos = 0
        for n in seqlens[b].tolist():
            n = int(n)
            if n <= 0:
                continue
            idx.extend(range(b * L + pos, b * L + pos + n))
            cu.append(cu[-1] + n)
            pos += n
    return torch.tensor(idx, dtype=torch.long), torch.tensor(cu, dtype=torch.int32)


def _norm(p, pre, x, ln):
    if ln:
        return F.layer_norm(x, x.shape[-1:], p[pre + 'weight'], p[pre + 'bias'], 1e-5)
    return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5) * p[pre + 'weight']   # TransformerFlashAttention.py:29-39


def cgpt_layer(p, pre, x, cfg, flags=None, bf16=True):
    """TransformerDecoder.forward (TransformerFlashAttention.py:104-121, DecoderLayer :76-85).  Eval mode / p = 0 unless
    `flags.dropout` carries a K.DropCounter: then the four dropout sites of every block (attention probabilities :67-70,
    post-attention :83, FFN hidden :52, post-FFN :84) draw counter-keyed masks in that order (oracle/kernels.py).
    MHA core: attention_alibi_varlen_ref (PARITY UNPINNED)."""
    B, L, D = x.shape
    H = cfg['nhead']
    hd = D // H
    if flags is not None and flags.seqlens is not None:
        idx, cu = seqlens_to_cu(flags.seqlens)
    else:
        idx = torch.arange(B * L)
        cu = torch.arange(0, (B + 1) * L, L, dtype=torch.int32)
    t = x.reshape(B * L, D)[idx]
    slopes = K.alibi_slopes(H)
    dc = flags.dropout if (flags is not None and cfg.get('pdrop', 0.0) > 0.0) else None
    pd = cfg.get('pdrop', 0.0) if dc is not None else 0.0
    drop = (lambda a: K.dropout_ref(a, pd, *dc.next())) if dc is not None else (lambda a: a)
    for i in range(cfg['nlayer']):
        lp = f'{pre}decoder_layers.{i}.'
        h = _norm(p, lp + 'mha_norm.', t, cfg['ln'])
        cast = (lambda a: a.to(torch.bfloat16).float()) if bf16 else (lambda a: a)
        qkv = cast(F.linear(cast(h), cast(p[lp + 'mha.Wqkv.weight']), cast(p[lp + 'mha.Wqkv.bias'])))
        qkv = qkv.view(-1, 3, H, hd)
        sd, of = dc.next() if dc is not None else (0, 0)
        a = cast(K.attention_alibi_varlen_ref(qkv[:, 0], qkv[:, 1], qkv[:, 2], cu, slopes, None, pd, sd, of, p_bf16=bf16))
        a = cast(F.linear(a.reshape(-1, D), cast(p[lp + 'mha.out_proj.weight']), cast(p[lp + 'mha.out_proj.bias'])))
        t = drop(a) + t                                             # :83
        h = _norm(p, lp + 'ffn_norm.', t, cfg['ln'])
        h = F.linear(drop(F.gelu(F.linear(h, p[lp + 'ffn.fc1.weight'], p[lp + 'ffn.fc1.bias']))),
                     p[lp + 'ffn.fc2.weight'], p[lp + 'ffn.fc2.bias'])
        t = drop(h) + t                                             # :84
    t = _norm(p, pre + 'output_ln.', t, cfg['ln'])
    t = F.linear(t, p[pre + 'output_fc.weight'], p[pre + 'output_fc.bias'])
    out = torch.zeros(B * L, D, dtype=t.dtype)
    out[idx] = t                                                    # pad_input, :120
    return out.view(B, L, D)


# ------------------------------------------------------------------------------------------------
# RNNBase / ContextualModel / policy / value
# ------------------------------------------------------------------------------------------------
def rnn_base_forward(p: Dict[str, torch.Tensor], spec: dict, x, flags: Optional[Flags] = None,
                     desire_ndim=None, smamba_semantics='gpu', gru_impl='ref'):
    """RNNBase.meta_forward (offpolicy_rnn/models/rnn_base.py:397-472).  spec = dict(layer_type, activation)."""
    for ind, (lt, act) in enumerate(zip(spec['layer_type'], spec['activation'])):
        pre = f'layer_list.{ind}.'
        c = parse_layer_id(lt)
        if c['kind'] == 'fc':
            x = F.linear(x, p[pre + 'weight'], p[pre + 'bias'])
        elif c['kind'] == 'efc':
            x = K.ensemble_linear_ref(x, p[pre + 'weight'], p.get(pre + 'bias'), desire_ndim)
        elif c['kind'] == 'gru':
            x = (gru_layer if gru_impl == 'ref' else gru_layer_aten)(p, pre, x, flags)
        elif c['kind'] == 'gilr':
            x = gilr_layer(p, pre, x, flags)
        elif c['kind'] == 'lru':
            x = lru_layer(p, pre, x, flags)
        elif c['kind'] == 'smamba':
            x = smamba_layer(p, pre, x, c, flags, smamba_semantics)
        elif c['kind'] == 'gilr_lstm':
            x = gilr_lstm_layer(p, pre, x, flags)[0]
        elif c['kind'] == 'conv1d':
            x = conv1d_layer(p, pre, x, flags)[0]
        elif c['kind'] == 'mamba':
            x = s6_mamba_layer(p, pre, x, c, flags)[0]
        elif c['kind'] == 'cgpt':
            x = cgpt_layer(p, pre, x, c, flags)
        if '+' in act:                                              # rnn_base.py:250-258, 461-467
            norm, name = act.split('+')
            apre = f'activation_list.{ind}.0.'
            if norm.startswith('eln'):
                xt = x.transpose(-2, 0)
                xt = F.layer_norm(xt, p[apre + 'weight'].shape, p[apre + 'weight'], p[apre + 'bias'])
                x = xt.transpose(-2, 0)
            else:
                x = F.layer_norm(x, x.shape[-1:], p[apre + 'weight'], p[apre + 'bias'])
            x = ACT[name](x)
        else:
            x = ACT[act](x)
    return x


def embedding_input(p, cfg, state, lst_state, lst_action, reward):
    """get_embedding_input (contextual_sac_value.py:90-99 / contextual_sac_policy_single_head.py:81-90)."""
    def enc(name, x):
        if cfg['separate_encoder']:
            return F.linear(x, p[name]['weight'], p[name]['bias'])
        return x
    parts = [enc('state_encoder', state)]
    if cfg['last_state_input']:
        parts.append(enc('last_obs_encoder', lst_state))
    if cfg['last_action_input']:
        parts.append(enc('last_act_encoder', lst_action))
    if cfg['reward_input']:
        parts.append(enc('reward_encoder', reward))
    return torch.cat(parts, dim=-1)


def _spec(cfg, which):
    return dict(layer_type=cfg[f'{which}_layer_type'], activation=cfg[f'{which}_activations'])


def categorical_head(out):
    """ContextualSACDiscretePolicy.process_model_out (contextual_sac_discrete_policy.py:111-125): softmax with a 0.01
    floor, through torch.distributions.Categorical (which renormalises).  -> (mode, sample, log-probs of all actions)."""
    probs = (out - torch.max(out, dim=-1, keepdim=True).values).exp()
    probs = probs / probs.sum(dim=-1, keepdim=True)
    probs = probs + 0.01
    probs = probs / probs.sum(dim=-1, keepdim=True)
    dist = torch.distributions.Categorical(probs=probs)
    return dist.mode.unsqueeze(-1), dist.sample().unsqueeze(-1), torch.log(dist.probs)


def policy_forward(p, cfg, state, lst_state, lst_action, flags=None, reward=None, noise=None,
                   algo='sac', sample_std=0.1, **kw):
    """ContextualSACPolicySingleHead.forward (contextual_sac_policy_single_head.py:92-107) and
    ContextualTD3Policy.forward (contextual_td3_policy.py:18-36).  `noise` replaces torch.randn_like."""
    emb_in = embedding_input(p, cfg, state, lst_state, lst_action, reward)
    emb = rnn_base_forward(p['embedding_model'], _spec(cfg, 'embedding'), emb_in, flags, **kw)
    uni_in = state
    if cfg['uni_model_input_mapping_dim'] > 0:                      # contextual_model.py:32-34,102
        m = p['uni_input_mapping_network']
        uni_in = ACT[cfg['embedding_activations'][-1]](F.linear(state, m['layer_list.0.weight'], m['layer_list.0.bias']))
    uni_act = list(cfg['uni_model_activations'][:-1]) + ['linear']  # contextual_sac_policy_single_head.py:20-21
    out = rnn_base_forward(p['universal_model'], dict(layer_type=cfg['uni_model_layer_type'], activation=uni_act),
                           torch.cat((uni_in, emb), dim=-1), flags, **kw)
    if cfg.get('discrete'):                                         # contextual_sac_discrete_policy.py:104-125
        mean, sample, logp = categorical_head(out)
        return mean, emb, sample, logp
    if algo == 'td3':
        mean = torch.tanh(out)
        noise = torch.randn_like(out) if noise is None else noise
        sample = torch.clamp(mean + noise * sample_std, -1, 1)
        return mean, emb, sample, torch.zeros_like(sample)
    logstd, mu = out.chunk(2, dim=-1)                               # :105 (logstd first)
    noise = torch.randn_like(mu) if noise is None else noise
    mean, sample, logp = K.tanh_gaussian_ref(mu, logstd, noise)
    return mean, emb, sample, logp


def rnn_base_step(p, spec, x, hidden):
    """RNNBase.meta_forward on ONE token with carried state (rnn_base.py:405-478): x [B, D], hidden = list of [B, hidden_k]
    per recurrent layer -> (y [B, D'], new hidden list)."""
    new, k = [], 0
    for ind, (lid, act) in enumerate(zip(spec['layer_type'], spec['activation'])):
        pre = f'layer_list.{ind}.'
        c = parse_layer_id(lid)
        if c['kind'] == 'fc':
            x = F.linear(x, p[pre + 'weight'], p[pre + 'bias'])
        elif c['kind'] == 'efc':
            x = K.ensemble_linear_ref(x, p[pre + 'weight'], p.get(pre + 'bias'), None)
        else:
            y, h = rollout_layer(p, lid, x.unsqueeze(1), hidden[k], pre)
            x = y[:, 0]
            new.append(h)
            k += 1
        assert '+' not in act
        x = ACT[act](x)
    return x, new


def policy_step(p, cfg, state, lst_state, lst_action, hidden, reward=None, noise=None, algo='sac', sample_std=0.1):
    """One rollout step of the policy (algorithm/sac.py:319-326 -> contextual_sac_policy_single_head.py:92-107 /
    contextual_td3_policy.py:18-36 on a single token).  hidden: recurrent states of the embedding stack.
    -> (mean, sample, logp, new hidden)."""
    emb_in = embedding_input(p, cfg, state, lst_state, lst_action, reward)
    emb, hidden = rnn_base_step(p['embedding_model'], _spec(cfg, 'embedding'), emb_in, hidden)
    uni_in = state
    if cfg['uni_model_input_mapping_dim'] > 0:
        m = p['uni_input_mapping_network']
        uni_in = ACT[cfg['embedding_activations'][-1]](F.linear(state, m['layer_list.0.weight'], m['layer_list.0.bias']))
    uni_act = list(cfg['uni_model_activations'][:-1]) + ['linear']
    out, _ = rnn_base_step(p['universal_model'], dict(layer_type=cfg['uni_model_layer_type'], activation=uni_act),
                           torch.cat((uni_in, emb), dim=-1), [])
    if algo == 'td3':
        mean = torch.tanh(out)
        noise = torch.randn_like(out) if noise is None else noise
        return mean, torch.clamp(mean + noise * sample_std, -1, 1), torch.zeros_like(mean), hidden
    logstd, mu = out.chunk(2, dim=-1)
    noise = torch.randn_like(mu) if noise is None else noise
    mean, sample, logp = K.tanh_gaussian_ref(mu, logstd, noise)
    return mean, sample, logp, hidden


def value_forward(p, cfg, state, lst_state, lst_action, action, flags=None, reward=None,
                  detach_embedding=False, desire_ndim=4, **kw):
    """ContextualSACValue.forward (contextual_sac_value.py:101-119).  Returns (Q [E,B,L,1], embedding)."""
    emb_in = embedding_input(p, cfg, state, lst_state, lst_action, reward)
    emb = rnn_base_forward(p['embedding_model'], _spec(cfg, 'embedding'), emb_in, flags, **kw)
    if detach_embedding:
        emb = emb.detach()                                          # contextual_model.py:70-71
    if cfg.get('discrete'):                                         # contextual_sac_discrete_value.py:99-110: phi_s(state) only
        if cfg['separate_encoder'] and cfg['uni_model_input_mapping_dim'] > 0:
            sa = ACT[cfg['embedding_activations'][-1]](F.linear(state, p['state_input_encoder_q']['weight'],
                                                                p['state_input_encoder_q']['bias']))
        else:
            sa = state
    elif cfg['separate_encoder'] and cfg['uni_model_input_mapping_dim'] > 0:
        sa = torch.cat((F.linear(state, p['state_input_encoder_q']['weight'], p['state_input_encoder_q']['bias']),
                        F.linear(action, p['action_input_encoder_q']['weight'], p['action_input_encoder_q']['bias'])), -1)
        sa = ACT[cfg['embedding_activations'][-1]](sa)              # contextual_sac_value.py:106-107
    else:
        sa = torch.cat((state, action), dim=-1)
    q = rnn_base_forward(p['universal_model'], _spec(cfg, 'uni_model'), torch.cat((sa, emb), dim=-1), flags,
                         desire_ndim=desire_ndim, **kw)
    return q, emb


# ------------------------------------------------------------------------------------------------
# parameter construction (shapes + init distributions of the reference; used by the timed baseline
# and by tests that do not load a golden state dict)
# ------------------------------------------------------------------------------------------------
def _xavier(out_f, in_f):
    w = torch.empty(out_f, in_f)
    torch.nn.init.xavier_uniform_(w)
    return w


def _default_linear(out_f, in_f, bias=True):
    lin = torch.nn.Linear(in_f, out_f, bias=bias)
    d = {'weight': lin.weight.detach().clone()}
    if bias:
        d['bias'] = lin.bias.detach().clone()
    return d


def _ff_init(p, pre, d):
    for n in ('w_1', 'w_2'):
        lin = _default_linear(d, d)
        p[pre + n + '.weight'], p[pre + n + '.bias'] = lin['weight'], lin['bias']
    p[pre + 'layer_norm.weight'] = torch.ones(d)
    p[pre + 'layer_norm.bias'] = torch.zeros(d)


def init_rnn_base(in_dim, out_dim, hidden, activation, layer_type) -> Dict[str, torch.Tensor]:
    """Shapes/inits of RNNBase.__init__ + xavier_initialize_weights (rnn_base.py:100-354)."""
    p = {}
    last = in_dim
    for ind, item in enumerate(list(hidden) + [out_dim]):
        pre = f'layer_list.{ind}.'
        c = parse_layer_id(layer_type[ind])
        k = c['kind']
        if k == 'fc':
            p[pre + 'weight'] = _xavier(item, last)
            p[pre + 'bias'] = torch.zeros(item)
        elif k == 'efc':
            E = c['ensemble']
            p[pre + 'weight'] = torch.stack([_xavier(item, last).t() for _ in range(E)])    # rnn_base.py:273-277
            p[pre + 'bias'] = torch.zeros(E, 1, item)
        elif k == 'gru':
            p[pre + 'weight_ih_l0'] = _xavier(3 * item, last)
            p[pre + 'weight_hh_l0'] = _xavier(3 * item, item)
            p[pre + 'bias_ih_l0'] = torch.zeros(3 * item)
            p[pre + 'bias_hh_l0'] = torch.zeros(3 * item)
        elif k == 'gilr':
            p[pre + 'in_proj.weight'] = torch.stack([_xavier(item, last).t() for _ in range(2)])
            p[pre + 'in_proj.bias'] = torch.zeros(2, 1, item)
            p[pre + 'out_proj.weight'] = _xavier(item, item)
            p[pre + 'out_proj.bias'] = torch.zeros(item)
            p[pre + 'layer_norm.weight'] = torch.ones(item)
            p[pre + 'layer_norm.bias'] = torch.zeros(item)
            _ff_init(p, pre + 'ff.', item)
        elif k == 'lru':
            p[pre + 'in_proj.weight'] = torch.stack([_xavier(item, last).t() for _ in range(3)])
            p[pre + 'in_proj.bias'] = torch.zeros(3, 1, item)
            p[pre + 'middle_proj.weight'] = torch.stack([_xavier(item, item).t() for _ in range(2)])
            p[pre + 'middle_proj.bias'] = torch.zeros(2, 1, item)
            u1, u2 = torch.rand(item), torch.rand(item)             # lru.py:50-68
            nu_log = torch.log(-0.5 * torch.log(u1 * (0.999 ** 2 - 0.9 ** 2) + 0.9 ** 2))
            theta_log = torch.log(u2 * math.pi * 2)
            gamma_log = torch.log(torch.sqrt(1 - torch.exp(-torch.exp(nu_log)) ** 2))
            p[pre + 'params_log'] = torch.vstack((nu_log, theta_log, gamma_log))
            _ff_init(p, pre + 'ff.', item)
        elif k == 'smamba':
            assert last == item
            D, N, Kc = item, c['d_state'], c['d_conv']
            Di, R = 2 * D, math.ceil(D / 16)
            for b in range(c['block_num']):
                bp = f'{pre}layers.{b}.'
                mp = bp + 'mixer.'
                p[mp + 'in_proj.weight'] = _default_linear(2 * Di, D, False)['weight']
                conv = torch.nn.Conv1d(Di, Di, Kc, groups=Di, padding=Kc - 1)
                p[mp + 'conv1d.weight'] = conv.weight.detach().clone()
                p[mp + 'conv1d.bias'] = conv.bias.detach().clone()
                p[mp + 'x_proj.weight'] = _default_linear(R + 2 * N, Di, False)['weight']
                std = R ** -0.5
                p[mp + 'dt_proj.weight'] = torch.empty(Di, R).uniform_(-std, std)           # mamba.py:94-98
                dt = torch.exp(torch.rand(Di) * (math.log(0.1) - math.log(0.001)) + math.log(0.001)).clamp(min=1e-4)
                p[mp + 'dt_proj.bias'] = dt + torch.log(-torch.expm1(-dt))                  # mamba.py:103-110
                p[mp + 'A_log'] = torch.log(torch.arange(1, N + 1, dtype=torch.float32)).repeat(Di, 1)
                p[mp + 'D'] = torch.ones(Di)
                w = torch.empty(D, Di)
                torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
                p[mp + 'out_proj.weight'] = w / math.sqrt(c['block_num'])                   # mamba.py:344-352
                p[bp + 'norm.weight'] = torch.ones(D)
                if not c['rms_norm']:
                    p[bp + 'norm.bias'] = torch.zeros(D)
            if c['use_ff']:
                _ff_init(p, pre + 'head.', D)
            else:
                p[pre + 'head.weight'] = _default_linear(D, D, False)['weight']
                p[pre + 'norm_f.weight'] = torch.ones(D)
                if not c['rms_norm']:
                    p[pre + 'norm_f.bias'] = torch.zeros(D)
        elif k == 'gilr_lstm':                                      # gilr_lstm.py:23-31, rnn_base.py:312-320
            p[pre + 'in_proj.weight'] = torch.stack([_xavier(item, last).t() for _ in range(2)])
            p[pre + 'in_proj.bias'] = torch.zeros(2, 1, item)
            p[pre + 'middle_proj.weight'] = torch.stack([_xavier(item, item).t() for _ in range(4)])
            p[pre + 'middle_proj.bias'] = torch.zeros(4, 1, item)
            p[pre + 'out_proj.weight'] = _xavier(item, item)
            p[pre + 'out_proj.bias'] = torch.zeros(item)
            p[pre + 'layer_norm.weight'] = torch.ones(item)
            p[pre + 'layer_norm.bias'] = torch.zeros(item)
        elif k == 'conv1d':                                         # conv1d.py:12-24 (torch default inits)
            conv = torch.nn.Conv1d(item, item, c['d_conv'], groups=item)
            p[pre + 'conv1d.weight'] = conv.weight.detach().clone()
            p[pre + 'conv1d.bias'] = conv.bias.detach().clone()
            _ff_init(p, pre + 'ff.', item)
        elif k == 'mamba':                                          # s6/mamba.py:69-128
            assert last == item
            D, N, Kc = item, c['d_state'], c['d_conv']
            Di, R = 2 * D, math.ceil(D / 16)
            mp = pre + 'mixer.'
            p[mp + 'in_proj.weight'] = _default_linear(2 * Di, D, False)['weight']
            conv = torch.nn.Conv1d(Di, Di, Kc, groups=Di)
            p[mp + 'conv1d.weight'] = conv.weight.detach().clone()
            p[mp + 'conv1d.bias'] = conv.bias.detach().clone()
            p[mp + 'x_proj.weight'] = _default_linear(R + 2 * N, Di, False)['weight']
            std = R ** -0.5
            p[mp + 'dt_proj.weight'] = torch.empty(Di, R).uniform_(-std, std)
            dt = torch.exp(torch.rand(Di) * (math.log(0.1) - math.log(0.001)) + math.log(0.001)).clamp(min=1e-4)
            p[mp + 'dt_proj.bias'] = dt + torch.log(-torch.expm1(-dt))
            p[mp + 'A_log'] = torch.log(torch.arange(1, N + 1, dtype=torch.float32)).repeat(Di, 1)
            p[mp + 'D'] = torch.ones(Di)
            p[mp + 'out_proj.weight'] = _default_linear(D, Di, False)['weight']
            p[pre + 'norm.weight'] = torch.ones(D)
            if c['use_ff']:
                _ff_init(p, pre + 'ff.', D)
            else:
                p[pre + 'ff.weight'] = _default_linear(D, D, False)['weight']
                p[pre + 'norm_f.weight'] = torch.ones(D)
        elif k == 'cgpt':
            D = last
            for i in range(c['nlayer']):
                lp = f'{pre}decoder_layers.{i}.'
                for name, o, n in (('mha.Wqkv', 3 * D, D), ('mha.out_proj', D, D), ('ffn.fc1', 4 * D, D), ('ffn.fc2', D, 4 * D)):
                    lin = _default_linear(o, n)
                    p[lp + name + '.weight'], p[lp + name + '.bias'] = lin['weight'], lin['bias']
                for nm in ('mha_norm', 'ffn_norm'):
                    p[lp + nm + '.weight'] = torch.ones(D)
                    if c['ln']:
                        p[lp + nm + '.bias'] = torch.zeros(D)
            p[pre + 'output_ln.weight'] = torch.ones(D)
            if c['ln']:
                p[pre + 'output_ln.bias'] = torch.zeros(D)
            lin = _default_linear(D, D)
            p[pre + 'output_fc.weight'], p[pre + 'output_fc.bias'] = lin['weight'], lin['bias']
        act = activation[ind]
        if '+' in act:
            norm = act.split('+')[0]
            shape = [int(norm.split('-')[-1]), item] if norm.startswith('eln') else [item]
            p[f'activation_list.{ind}.0.weight'] = torch.ones(shape)
            p[f'activation_list.{ind}.0.bias'] = torch.zeros(shape)
        last = item
    return p


def init_model(cfg, kind: str) -> Dict[str, Dict[str, torch.Tensor]]:
    """Module set + ordering of ContextualSACPolicySingleHead / ContextualSACValue."""
    sd, ad = cfg['state_dim'], cfg['action_dim']
    basic = 128
    if cfg['separate_encoder']:
        cum = basic * (1 + int(cfg['last_action_input']) + int(cfg['last_state_input']) + int(cfg['reward_input']))
    else:
        cum = sd + (ad if cfg['last_action_input'] else 0) + (sd if cfg['last_state_input'] else 0) + int(cfg['reward_input'])
    p = {}
    p['embedding_model'] = init_rnn_base(cum, cfg['embedding_size'], cfg['embedding_hidden'],
                                         cfg['embedding_activations'], cfg['embedding_layer_type'])
    mapping = cfg['uni_model_input_mapping_dim']
    if kind == 'policy':
        uni_in = sd if mapping == 0 else mapping
        out = ad * 2 if cfg.get('algo', 'sac') == 'sac' and not cfg.get('discrete') else ad
        p['universal_model'] = init_rnn_base(cfg['embedding_size'] + uni_in, out, cfg['uni_model_hidden'],
                                             cfg['uni_model_activations'], cfg['uni_model_layer_type'])
        if mapping > 0:
            p['uni_input_mapping_network'] = init_rnn_base(sd, mapping, [], [cfg['embedding_activations'][-1]], ['fc'])
    else:
        uni_in = sd if cfg.get('discrete') else sd + ad
        if mapping > 0 and cfg['separate_encoder']:
            uni_in = mapping if cfg.get('discrete') else mapping * 2
        p['universal_model'] = init_rnn_base(cfg['embedding_size'] + uni_in, ad if cfg.get('discrete') else 1, cfg['uni_model_hidden'],
                                             cfg['uni_model_activations'], cfg['uni_model_layer_type'])
    if cfg['separate_encoder']:
        p['state_encoder'] = _default_linear(basic, sd)
        if cfg['last_action_input']:
            p['last_act_encoder'] = _default_linear(basic, ad)
        if cfg['last_state_input']:
            p['last_obs_encoder'] = _default_linear(basic, sd)
        if cfg['reward_input']:
            p['reward_encoder'] = _default_linear(basic, 1)
        if kind == 'value' and mapping > 0:
            p['state_input_encoder_q'] = _default_linear(mapping, sd)
            if not cfg.get('discrete'):
                p['action_input_encoder_q'] = _default_linear(mapping, ad)
    return p


def flat_params(p) -> List[torch.Tensor]:
    return [t for mod in p.values() for t in mod.values()]


def l2_norm_square(p) -> torch.Tensor:
    """ContextualModel.l2_norm_square (contextual_model.py:227-228): only RNNBase modules have the method,
    i.e. embedding_model, universal_model and uni_input_mapping_network - plain nn.Linear encoders are skipped."""
    tot = 0.0
    for name in ('embedding_model', 'universal_model', 'uni_input_mapping_network'):
        if name in p:
            tot = tot + sum(torch.sum(t ** 2) for t in p[name].values())
    return tot
