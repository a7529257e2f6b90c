"""Test-only kernel stand-ins: route `offpolicy_rnn.hip.ops` to the CPU oracle so that the HOST logic of the product
(buffers -> trainer orchestration -> optimizer / DP plumbing) can be exercised in the GPU-less container.

This is test infrastructure: the product never imports it, and on a GPU box the real HIP kernels run instead
(tests/test_hip_ops.py, tests/test_trainer_gpu.py).  Installed by the `oracle_ops` fixture in conftest.py."""
import torch

from oracle import kernels as K


def _flag(t, B, L):
    return None if t is None else t.reshape(B, L)


def selective_scan_tm(u, delta, A, Bm, Cm, D=None, z=None, delta_bias=None, start=None, delta_softplus=True, return_last_state=False):
    out, last = K.selective_scan_ref(u, delta, A, Bm, Cm, D, z, delta_bias, _flag(start, *u.shape[:2]), delta_softplus)
    return (out, last) if return_last_state else out


def causal_conv1d_fn(x, weight, bias=None, mask=None, activation=True):
    return K.causal_conv1d_silu_ref(x, weight.reshape(x.shape[-1], -1), bias, _flag(mask, *x.shape[:2]), activation)


def mamba_inner_fn(x, in_w, conv_w, conv_b, xproj_w, dt_w, dt_b, A_log, D, out_w, mask=None, start=None):
    import torch.nn.functional as F
    Di, N = A_log.shape
    R = dt_w.shape[1]
    xz = F.linear(x, in_w)
    xc = causal_conv1d_fn(xz[..., :Di], conv_w, conv_b, mask, True)
    x_dbl = F.linear(xc, xproj_w)
    dt = F.linear(x_dbl[..., :R], dt_w)
    y = selective_scan_tm(xc, dt, -torch.exp(A_log.float()), x_dbl[..., R:R + N], x_dbl[..., R + N:], D, xz[..., Di:], dt_b, start, True)
    return F.linear(y, out_w)


def bias_act_(y2, bias2, rows_per_seg, act):
    with torch.no_grad():
        if bias2 is not None:
            nseg = bias2.shape[0]
            y2.view(nseg, -1, y2.shape[1]).add_(bias2.view(nseg, 1, -1))
        if act == 'elu':
            y2.copy_(torch.nn.functional.elu(y2))
    return y2


def bias_act_bwd(g2, a2, rows_per_seg, act, need_dbias):
    gy = g2
    if act == 'elu':
        gy = g2 * torch.where(a2 > 0, torch.ones_like(a2), a2 + 1)
    db = gy.reshape(-1, int(rows_per_seg), gy.shape[1]).sum(1) if need_dbias else None
    return gy.contiguous(), db


def ensemble_head_fwd_(y3, b2, w3, b3):
    with torch.no_grad():
        y3.copy_(torch.nn.functional.elu(y3 + b2.unsqueeze(1)))
        q = (y3 * w3.unsqueeze(1)).sum(-1)
        return q if b3 is None else q + b3.view(-1, 1)


def ensemble_head_bwd(gq, a3, w3):
    gy = gq.unsqueeze(-1) * w3.unsqueeze(1) * torch.where(a3 > 0, torch.ones_like(a3), a3 + 1)
    return gy, gy.sum(1), (a3 * gq.unsqueeze(-1)).sum(1)


def linear_act(x, weight, bias, act, dest=None):
    y = torch.nn.functional.linear(x, weight, bias)
    return torch.nn.functional.elu(y) if act == 'elu' else y


def gemm_f32(A, B, a_kcontig=True, b_kcontig=True, bias=None, act=None, out=None, split=None, amax_a=None, amax_b=None, amax_out=None):
    """CPU stand-in for the product's one GEMM entry (hip/ops.py `gemm_f32`: A [M, K] or [K, M], B [N, K] or [K, N], optional leading
    ensemble axis, bias / ELU / accumulate tails)."""
    a = A if a_kcontig else A.transpose(-1, -2)
    b = B.transpose(-1, -2) if b_kcontig else B
    y = torch.matmul(a, b)
    if bias is not None:
        y = y + (bias.reshape(y.shape[0], 1, -1) if y.dim() == 3 else bias.reshape(1, -1))
    if act == 'elu':
        y = torch.nn.functional.elu(y)
    elif act == 'softplus':
        y = torch.nn.functional.softplus(y)
    elif act == 'accumulate':
        y = out + y
    elif act not in (None, 'linear'):
        raise KeyError(act)
    if out is not None:
        out.copy_(y)
        return out
    return y


def linear_bf16(x, weight, bias, out_dtype=torch.bfloat16, round_out=False):
    """The bf16-autocast projection of the cgpt MHA as the reference's autocast graph spells it (casts around F.linear)."""
    bf = torch.bfloat16
    y = torch.nn.functional.linear(x.to(bf), weight.to(bf), None if bias is None else bias.to(bf))
    return y.to(out_dtype)


def _norm(rms):
    def fn(x, weight, bias, residual=None, eps=1e-6, prenorm=False, residual_in_fp32=False, act=None):
        y, res = K.add_layernorm_ref(x, residual, weight, bias, eps, rms)
        if act == 'elu':
            y = torch.nn.functional.elu(y)
        return (y, res) if prenorm else y
    return fn


def gilr_scan(v, f, start=None, h0=None, fuse_act=True):
    return K.linrec_real_ref(v, f, _flag(start, *v.shape[:2]), h0, fuse_act)[0]


def complex_scan(vr, vi, lam_re, lam_im, gamma=None, start=None, h0r=None, h0i=None):
    h0r = None if h0r is None else h0r.reshape(vr.shape[0], -1)
    h0i = None if h0i is None else h0i.reshape(vr.shape[0], -1)
    return K.linrec_complex_ref(vr, vi, lam_re, lam_im, _flag(start, *vr.shape[:2]), h0r, h0i, gamma)


def gilr_scan_members(u, start=None, h0=None, fuse_act=True):
    return gilr_scan(u[0], u[1], start, h0, fuse_act)


def lru_params(params_log):
    nu, theta, gamma = torch.exp(params_log)
    mag = torch.exp(-nu)
    return torch.stack((mag * torch.cos(theta), mag * torch.sin(theta), gamma))


def complex_scan_members(u, lam3, start=None, h0r=None, h0i=None):
    hr, hi = complex_scan(u[0], u[1], lam3[0], lam3[1], lam3[2], start, h0r, h0i)
    return torch.stack((hr, hi), dim=0), (u[2] if u.shape[0] == 3 else None)


class SubAddMembers:
    @staticmethod
    def apply(m, r):
        return m[0] - m[1] + r if r is not None else m[0] - m[1]


def gru_seq(gi, w_hh, b_hh, h0=None):
    return K.gru_seq_ref(gi, w_hh, b_hh, h0)


def attn_varlen(qkv, cu_seqlens, max_seqlen, slopes=None, scale=None, p_drop=0.0, seed=0, offset=0):
    out = K.attention_alibi_varlen_ref(qkv[:, 0], qkv[:, 1], qkv[:, 2], cu_seqlens, slopes, scale, p_drop, seed, offset, p_bf16=True)
    return out.to(torch.bfloat16)


_host_counter = K.DropCounter(0)


def dropout_counter(device):
    return _host_counter.next()


def counter_dropout(x, p_drop, seed=None, offset=None):
    if p_drop <= 0.0:
        return x
    if seed is None:
        seed, offset = dropout_counter(x.device)
    return K.dropout_ref(x, p_drop, seed, offset)


def tanh_gaussian(out2, noise):
    A = out2.shape[-1] // 2
    mean, samp, logp = K.tanh_gaussian_ref(out2[..., A:], out2[..., :A], noise)
    return mean.detach(), samp, logp


@torch.no_grad()
def sac_target(q, subset, next_logp, log_alpha, reward, done, mask, gamma, guard, stats=None, reduce_max=None, local_ext=None):
    idx = subset.long()
    v = q[idx].min(dim=0).values
    if next_logp is not None:
        v = v - log_alpha.exp() * next_logp
    if local_ext is not None:                       # one-collective data parallelism: guard as it stands, extrema handed out
        lo, hi = (guard[0].item(), guard[1].item()) if guard[2] != 0 else (-float('inf'), float('inf'))
        y = reward + (1 - done) * gamma * v.clamp(min=lo, max=hi)
        ym = y * mask
        local_ext.copy_(torch.stack((-v.min(), v.max(), -ym.min(), ym.max())))
        if stats is not None:
            stats[0], stats[1] = y.abs().max(), mask.sum()
        return y
    ext = torch.stack((-v.min(), v.max()))
    if reduce_max is not None:
        reduce_max(ext)
    if guard[2] == 0:
        guard[0], guard[1], guard[2] = -ext[0], ext[1], 1.0
    y = reward + (1 - done) * gamma * v.clamp(min=guard[0].item(), max=guard[1].item())
    ym = y * mask
    ext = torch.stack((-ym.min(), ym.max()))
    if reduce_max is not None:
        reduce_max(ext)
    bmin, bmax = -ext[0], ext[1]
    lo, hi = torch.minimum(guard[0], bmin), torch.maximum(guard[1], bmax)
    decay = guard[3]
    if decay < 1:
        lo, hi = decay * lo + (1 - decay) * bmin, decay * hi + (1 - decay) * bmax
    guard[0], guard[1] = lo, hi
    if stats is not None:
        stats[0], stats[1] = y.abs().max(), mask.sum()
    return y


@torch.no_grad()
def guard_apply_slots(slots, world, guard):
    e = slots.reshape(world, 4).max(dim=0).values
    if guard[2] == 0:
        guard[0], guard[1], guard[2] = -e[0], e[1], 1.0
    bmin, bmax = -e[2], e[3]
    lo, hi = torch.minimum(guard[0], bmin), torch.maximum(guard[1], bmax)
    decay = guard[3]
    if decay < 1:
        lo, hi = decay * lo + (1 - decay) * bmin, decay * hi + (1 - decay) * bmax
    guard[0], guard[1] = lo, hi


@torch.no_grad()
def soft_update_(target_flat, online_flat, tau):
    target_flat.copy_(K.soft_update_ref(target_flat, online_flat, tau))


@torch.no_grad()
def adamw_flat_(p, g, m, v, seg_end, seg_lr, seg_wd, step, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=None):
    gs = g if grad_scale is None else g * grad_scale
    a = 0
    for e, lr, wd in zip(seg_end.tolist(), seg_lr.tolist(), seg_wd.tolist()):
        pn, mn, vn = K.adamw_ref(p[a:e], gs[a:e], m[a:e], v[a:e], step, lr, beta1, beta2, eps, wd)
        p[a:e], m[a:e], v[a:e] = pn, mn, vn
        a = e


@torch.no_grad()
def sumsq(x, out=None):
    r = (x.float() ** 2).sum().reshape(1)
    if out is not None:
        out.copy_(r)
        return out
    return r


def install(monkeypatch):
    from offpolicy_rnn.hip import ops
    table = dict(ensemble_head_fwd_=ensemble_head_fwd_, ensemble_head_bwd=ensemble_head_bwd, bias_act_=bias_act_, bias_act_bwd=bias_act_bwd, linear_act=linear_act, gemm_f32=gemm_f32, linear_bf16=linear_bf16, gemm_bf16_ok=lambda x, w: True, mamba_inner_fn=mamba_inner_fn, selective_scan_tm=selective_scan_tm, causal_conv1d_fn=causal_conv1d_fn, layer_norm_fn=_norm(False),
                 rms_norm_fn=_norm(True), gilr_scan=gilr_scan, complex_scan=complex_scan, gilr_scan_members=gilr_scan_members, complex_scan_members=complex_scan_members, lru_params=lru_params, SubAddMembers=SubAddMembers, gru_seq=gru_seq, tanh_gaussian=tanh_gaussian, attn_varlen=attn_varlen, dropout_counter=dropout_counter, counter_dropout=counter_dropout,
                 sac_target=sac_target, guard_apply_slots=guard_apply_slots, soft_update_=soft_update_, adamw_flat_=adamw_flat_, sumsq=sumsq)
    for k, fn in table.items():
        monkeypatch.setattr(ops, k, fn)
    return ops
