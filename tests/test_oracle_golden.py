"""The CPU oracle against the golden vectors generated from the reference itself (pins the oracle)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN, load_golden, nested
from oracle import kernels as K
from oracle import network as NW
from oracle.buffer import OracleBuffer, Transition
from oracle.trainer import OracleTrainer, default_parameter

T = torch.from_numpy


def _push(buf, o, a, r, early_done):
    n = len(a)
    for t in range(n):
        buf.mem_push(Transition(
            state=o[t:t + 1], last_state=o[t - 1:t] if t > 0 else np.zeros((1, o.shape[1])),
            last_action=a[t - 1:t] if t > 0 else np.zeros((1, a.shape[1])), action=a[t:t + 1], next_state=o[t + 1:t + 2],
            reward=float(r[t]), logp=None, mask=1, start=(t == 0), done=(t == n - 1),
            reward_input=np.array([[r[t - 1] if t > 0 else 0.0]]), timeout=(t == n - 1) and not early_done))


def _synth(rs, n, obs, act):
    return rs.randn(n + 1, obs), np.tanh(rs.randn(n, act)), rs.randn(n)


@pytest.mark.parametrize('case', ['skip2_nonest', 'skip18_nest', 'skip2_nest_rmask', 'skip2_fixedT'])
def test_sample_trajs_layout(case):
    g = load_golden('sample_trajs.npz')
    cfg = g[f'{case}|cfg']
    hist, nest, rmask, bs, maxT = [int(v) for v in cfg[:5]]
    lens = [int(v) for v in cfg[5:]]
    buf = OracleBuffer(1000, maxT, additional_history_len=hist)
    rs = np.random.RandomState(7)
    for n in lens:
        o, a, r = _synth(rs, n, 4, 2)
        _push(buf, o, a, r, early_done=(n != 12))
    np.random.seed(123)
    res, total, valid, table = buf.sample_trajs(bs, randomize_mask=bool(rmask), valid_number_post_randomized=9,
                                                equalize_data_of_each_traj=True, nest_stack_trajs=bool(nest))
    assert total == int(g[f'{case}|total'])
    np.testing.assert_array_equal(valid, g[f'{case}|valid'])
    np.testing.assert_array_equal(table, g[f'{case}|table'])
    for f in res._fields:
        if getattr(res, f) is not None:
            np.testing.assert_array_equal(getattr(res, f), g[f'{case}|{f}'], err_msg=f)


@pytest.mark.parametrize('case', ['n16', 'n32', 'n64'])
def test_selective_scan_ref_fwd_bwd(case):
    g = load_golden('selective_scan.npz')
    tm = lambda k: T(g[f'{case}|{k}']).transpose(1, 2).contiguous()      # reference (B, D, L) -> token-major
    u, delta, z, Bm, Cm = [tm(k).requires_grad_(True) for k in ('u', 'delta', 'z', 'Bm', 'Cm')]
    A, D, db = [T(g[f'{case}|{k}']).requires_grad_(True) for k in ('A', 'D', 'delta_bias')]
    start = T(g[f'{case}|start'])
    out, last = K.selective_scan_ref(u, delta, A, Bm, Cm, D, z, db, start, True)
    np.testing.assert_allclose(out.detach().transpose(1, 2), g[f'{case}|out'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(last.detach(), g[f'{case}|last_state'], rtol=1e-5, atol=1e-5)
    (out * tm('dout')).sum().backward()
    for k, t in dict(u=u, delta=delta, z=z, Bm=Bm, Cm=Cm).items():
        np.testing.assert_allclose(t.grad.transpose(1, 2), g[f'{case}|d{k}'], rtol=2e-4, atol=2e-4, err_msg=k)
    for k, t in dict(A=A, D=D, delta_bias=db).items():
        np.testing.assert_allclose(t.grad, g[f'{case}|d{k}'], rtol=2e-4, atol=2e-4, err_msg=k)


def _layer_params(g, tag):
    pre = f'{tag}|p|'
    return {k[len(pre):]: T(np.array(v)).requires_grad_(True) for k, v in g.items() if k.startswith(pre)}


@pytest.mark.parametrize('lid', ['gru', 'gilr', 'lru', 'gilr_lstm', 'conv1d_5', 'mamba_s8_c3', 'mamba_s4_c5_noff'])
def test_layer_fwd_bwd(lid):
    g = load_golden('layers.npz')
    p = _layer_params(g, lid)
    x = T(g['x']).requires_grad_(True)
    flags = NW.Flags(T(g['start']), T(g['mask']))
    y = NW.rnn_base_forward(p, dict(layer_type=[lid], activation=['linear']), x, flags)
    np.testing.assert_allclose(y.detach(), g[f'{lid}|y'], rtol=1e-4, atol=2e-5)
    (y * T(g['w'])).sum().backward()
    np.testing.assert_allclose(x.grad, g[f'{lid}|dx'], rtol=1e-3, atol=1e-4)
    for k, v in g.items():
        if k.startswith(f'{lid}|g|'):
            name = k.split('|')[-1]
            np.testing.assert_allclose(p[name].grad, v, rtol=2e-3, atol=2e-4, err_msg=name)


@pytest.mark.parametrize('lid', ['smamba_s8_c6_b2_nln', 'smamba_s16_c4_b1', 'smamba_s8_c5_b1_ff'])
def test_smamba_layer_both_semantics(lid):
    g = load_golden('layers.npz')
    p = _layer_params(g, lid)
    x = T(g['x']).requires_grad_(True)
    flags = NW.Flags(T(g['start']), T(g['mask']))
    spec = dict(layer_type=[lid], activation=['linear'])
    y = NW.rnn_base_forward(p, spec, x, flags, smamba_semantics='gpu')
    np.testing.assert_allclose(y.detach(), g[f'{lid}|y_seq'], rtol=1e-4, atol=2e-5)
    (y * T(g['w'])).sum().backward()
    np.testing.assert_allclose(x.grad, g[f'{lid}|dx_seq'], rtol=1e-3, atol=1e-4)
    for k, v in g.items():
        if k.startswith(f'{lid}|gseq|'):
            name = k.split('|')[-1]
            np.testing.assert_allclose(p[name].grad, v, rtol=2e-3, atol=3e-4, err_msg=name)
    y2 = NW.rnn_base_forward(p, spec, T(g['x']), flags, smamba_semantics='cpu_step')
    np.testing.assert_allclose(y2.detach(), g[f'{lid}|y_step'], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize('lid', ['gru', 'gilr', 'lru', 'smamba_s8_c6_b2_nln', 'smamba_s16_c4_b1', 'smamba_s8_c5_b1_ff', 'gilr_lstm',
                                 'conv1d_5', 'mamba_s8_c3', 'mamba_s4_c5_noff'])
def test_rollout_steps(lid):
    """One-token steps with a carried (random) hidden state, as recorded from the reference's own modules."""
    g = load_golden('rollout.npz')
    p = {k: v.detach() for k, v in _layer_params(g, lid).items()}
    y, hT = NW.rollout_layer(p, lid, T(g['x']), T(g[f'{lid}|h0'])[0])
    np.testing.assert_allclose(y, g[f'{lid}|y'], rtol=1e-4, atol=2e-5)
    # (conv1d / mamba return the state as (B, 1, X) - conv1d.py:45, s6/mamba.py:185-188 - same memory order as (1, B, X))
    np.testing.assert_allclose(hT, g[f'{lid}|hT'].reshape(hT.shape), rtol=1e-4, atol=2e-5)


def test_layer_id_table():
    table = json.load(open(os.path.join(GOLDEN, 'layer_ids.json')))
    for lid, e in table.items():
        assert NW.hidden_size_of(lid, 32, 32) == e['hidden'], lid
        p = NW.init_rnn_base(32, 32, [], ['linear'], [lid])
        assert sum(t.numel() for t in p.values()) == e['nparam'], lid
        if lid.startswith('smamba'):
            c = NW.parse_layer_id(lid)
            for k in ('d_conv', 'd_state', 'block_num', 'rms_norm', 'use_ff'):
                assert c[k] == e[k], (lid, k)


META = json.load(open(os.path.join(GOLDEN, 'train_meta.json')))


def _trainer(name):
    m = META[name]
    g = load_golden(f'train_{name}.npz')
    par = default_parameter(rnn=m['rnn'], D=32, algo=m['algo'], sac_batch_size=m['sac_batch_size'],
                            policy_embedding_dim=16, value_embedding_dim=16, policy_uni_model_input_mapping_dim=16,
                            value_uni_model_input_mapping_dim=16, max_buffer_transition_num=5000)
    sem = 'cpu_step' if m['rnn'].startswith('smamba') else 'gpu'    # the goldens ran the reference on CPU tensors
    tr = OracleTrainer(par, 5, 3, 12, smamba_semantics=sem, policy_state=nested(g, 'policy0|'),
                       value_state=nested(g, 'value0|'))
    rs = np.random.RandomState(9)
    for n in m['lens']:
        o, a, r = _synth(rs, n, 5, 3)
        _push(tr.buffer, o, a, r, early_done=(n != 12))
    return tr, g, m


@pytest.mark.parametrize('name', list(META))
def test_policy_value_forward(name):
    tr, g, m = _trainer(name)
    assert tr.buffer.skip == m['skip_len'] + 1 and tr.nest == m['nest']
    f = lambda k: T(g[k])
    flags = NW.Flags(f('fw_start'), f('fw_valid'))
    mean, emb, samp, logp = NW.policy_forward(tr.policy, tr.pcfg, f('fw_state'), f('fw_last_state'), f('fw_last_action'),
                                              flags, None, noise=f('fw_noise'), algo=tr.algo, **tr.fw)
    q, qemb = NW.value_forward(tr.value, tr.vcfg, f('fw_state'), f('fw_last_state'), f('fw_last_action'), f('fw_action'),
                               flags, None, **tr.fw)
    for k, v in dict(fw_mean=mean, fw_emb=emb, fw_sample=samp, fw_logp=logp, fw_q=q, fw_qemb=qemb).items():
        np.testing.assert_allclose(v.detach(), g[k], rtol=1e-4, atol=2e-5, err_msg=k)


@pytest.mark.parametrize('name', list(META))
def test_train_one_batch_three_updates(name):
    tr, g, m = _trainer(name)
    torch.manual_seed(200)
    np.random.seed(200)
    for it in range(3):
        log = tr.train_one_batch()
        tr.grad_num += 1
        ref = m['logs'][it]
        for k, v in ref.items():
            got = log[k][0] if isinstance(log[k], tuple) else log[k]
            assert got == pytest.approx(v, rel=2e-3, abs=2e-4), (it, k, got, v)
    for pre, net in (('policy3|', tr.policy), ('value3|', tr.value), ('target3|', tr.target_value)):
        ref = nested(g, pre)
        for mname, d in ref.items():
            for k, v in d.items():
                np.testing.assert_allclose(net[mname][k].detach(), v, rtol=2e-3, atol=2e-5, err_msg=f'{pre}{mname}.{k}')
    np.testing.assert_allclose(tr.log_alpha.detach(), g['log_alpha3'], rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------------------------------------- discrete actions
DMETA = json.load(open(os.path.join(GOLDEN, 'train_discrete_meta.json')))


def push_discrete(buffer, Transition, rs, L, obs, act, T):
    """Same draws as generate_golden.gen_discrete_train: integer actions, one-hot last actions."""
    o, r = rs.randn(L + 1, obs), rs.randn(L)
    a = rs.randint(act, size=(L, 1)).astype(np.float64)
    oh = np.eye(act)[a[:, 0].astype(int)]
    for t in range(L):
        buffer.mem_push(Transition(
            state=o[t:t + 1], last_state=o[t - 1:t] if t > 0 else np.zeros((1, obs)),
            last_action=oh[t - 1:t] if t > 0 else np.zeros((1, act)), action=a[t:t + 1], next_state=o[t + 1:t + 2],
            reward=float(r[t]), logp=None, mask=1, start=(t == 0), done=(t == L - 1),
            reward_input=np.array([[r[t - 1] if t > 0 else 0.0]]), timeout=(t == L - 1) and L == T))


def discrete_trainer(name):
    from oracle.buffer import Transition
    m = DMETA[name]
    g = load_golden(f'train_{name}.npz')
    par = default_parameter(rnn=m['rnn'], D=32, algo='sac', sac_batch_size=m['sac_batch_size'], sac_alpha=m['sac_alpha'],
                            policy_embedding_dim=16, value_embedding_dim=16, policy_uni_model_input_mapping_dim=16,
                            value_uni_model_input_mapping_dim=16, max_buffer_transition_num=5000)
    tr = OracleTrainer(par, 5, m['n_actions'], 12, policy_state=nested(g, 'policy0|'), value_state=nested(g, 'value0|'), discrete=True)
    rs = np.random.RandomState(9)
    for n in m['lens']:
        push_discrete(tr.buffer, Transition, rs, n, 5, m['n_actions'], 12)
    return tr, g, m


@pytest.mark.parametrize('name', list(DMETA))
def test_discrete_policy_value_forward(name):
    tr, g, m = discrete_trainer(name)
    f = lambda k: T(g[k])
    flags = NW.Flags(f('fw_start'), f('fw_valid'))
    mean, emb, samp, logp = NW.policy_forward(tr.policy, tr.pcfg, f('fw_state'), f('fw_last_state'), f('fw_last_action'), flags, None, **tr.fw)
    q, qemb = NW.value_forward(tr.value, tr.vcfg, f('fw_state'), f('fw_last_state'), f('fw_last_action'), f('fw_action'), flags, None, **tr.fw)
    assert q.shape[-1] == m['n_actions'] and logp.shape[-1] == m['n_actions']
    for k, v in dict(fw_mean=mean.float(), fw_emb=emb, fw_logp=logp, fw_q=q, fw_qemb=qemb).items():
        np.testing.assert_allclose(v.detach(), g[k], rtol=1e-4, atol=2e-5, err_msg=k)


@pytest.mark.parametrize('name', list(DMETA))
def test_discrete_three_updates(name):
    tr, g, m = discrete_trainer(name)
    torch.manual_seed(200)
    np.random.seed(200)
    for it in range(3):
        log = tr.train_one_batch()
        tr.grad_num += 1
        for k, v in m['logs'][it].items():
            got = log[k][0] if isinstance(log[k], tuple) else log[k]
            assert got == pytest.approx(v, rel=2e-3, abs=2e-4), (it, k, got, v)
    for pre, net in (('policy3|', tr.policy), ('value3|', tr.value), ('target3|', tr.target_value)):
        for mname, d in nested(g, pre).items():
            for k, v in d.items():
                np.testing.assert_allclose(net[mname][k].detach(), v, rtol=2e-3, atol=2e-5, err_msg=f'{pre}{mname}.{k}')


# ---------------------------------------------------------------------------------------------- cgpt attention (unpinned)
def test_alibi_slopes_are_the_published_geometric_sequences():
    """flash_attn is un-vendored (parity unpinned): anchor the restatement on the published ALiBi definition
    (slopes 2^(-8 i / H) for power-of-two head counts; interleaved sequence otherwise)."""
    np.testing.assert_allclose(K.alibi_slopes(8), [2.0 ** -(i + 1) for i in range(8)], rtol=0, atol=0)
    np.testing.assert_allclose(K.alibi_slopes(4), [2.0 ** -(2 * (i + 1)) for i in range(4)], rtol=1e-7)
    s12 = K.alibi_slopes(12)
    assert len(s12) == 12 and np.allclose(s12[:8], [2.0 ** -(i + 1) for i in range(8)]) and np.allclose(s12[8:], [2.0 ** -(i + 0.5) for i in range(4)])


@pytest.mark.parametrize('H,d,lens', [(8, 32, [7, 1, 19]), (4, 16, [33])])
def test_attention_restatement_equals_torch_sdpa_with_alibi_bias(H, d, lens):
    """Independent implementation check of the oracle's packed causal + ALiBi attention: torch's own
    scaled_dot_product_attention with an explicit additive bias -slope_h (i - j) and a causal mask, per sequence."""
    g = torch.Generator().manual_seed(3)
    Tn = sum(lens)
    q, k, v = (torch.randn(Tn, H, d, generator=g) for _ in range(3))
    cu = torch.tensor(np.concatenate(([0], np.cumsum(lens))), dtype=torch.int32)
    slopes = K.alibi_slopes(H)
    got = K.attention_alibi_varlen_ref(q, k, v, cu, slopes)
    for s in range(len(lens)):
        a, b = int(cu[s]), int(cu[s + 1])
        n = b - a
        i, j = torch.arange(n)[:, None], torch.arange(n)[None, :]
        bias = -slopes[:, None, None] * (i - j).float()[None]
        bias = bias.masked_fill((j > i)[None], float('-inf'))
        ref = F.scaled_dot_product_attention(q[a:b].transpose(0, 1)[None], k[a:b].transpose(0, 1)[None], v[a:b].transpose(0, 1)[None],
                                             attn_mask=bias[None])[0].transpose(0, 1)
        np.testing.assert_allclose(got[a:b], ref, rtol=1e-5, atol=1e-6)
