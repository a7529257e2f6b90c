"""One-token rollout steps on the GPU: the step kernels against the CPU oracle, the layers
against vectors recorded from the reference's own T == 1 code paths (tests/golden/rollout.npz), KV-cache decoding against
the packed training kernel, and the hipGraph replay of a whole policy step against the eager step."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import kernels as K

pytestmark = pytest.mark.gpu
ROLLOUT_IDS = ['gru', 'gilr', 'lru', 'smamba_s8_c6_b2_nln', 'smamba_s16_c4_b1', 'smamba_s8_c5_b1_ff', 'gilr_lstm', 'conv1d_5',
               'mamba_s8_c3', 'mamba_s4_c5_noff']


@pytest.fixture(scope='module')
def ops():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from offpolicy_rnn.hip import ops as o
    return o


def T(a):
    return torch.from_numpy(np.array(a)).float()


# ------------------------------------------------------------------------------------------------ kernels vs oracle
@pytest.mark.parametrize('B,Di,N,Kw,R', [(1, 512, 32, 16, 16), (3, 96, 8, 4, 3), (2, 40, 5, 6, 2), (5, 64, 64, 2, 4)])
def test_mamba_step_kernels_vs_oracle(ops, B, Di, N, Kw, R):
    g = torch.Generator().manual_seed(B * 100 + Di)
    rn = lambda *s: torch.randn(*s, generator=g)
    total = Di * (Kw + N) + 8                                     # the layer's chunk sits inside a wider hidden row
    hidden = rn(B, total)
    xz = rn(B, 2 * Di)
    conv_w, conv_b = rn(Di, 1, Kw) * 0.4, rn(Di) * 0.1
    xproj_w, dt_w, dt_b = rn(R + 2 * N, Di) * Di ** -0.5, rn(Di, R) * R ** -0.5, rn(Di) * 0.5 - 2.0
    A_log, D = torch.log(torch.arange(1, N + 1, dtype=torch.float32)).repeat(Di, 1) + rn(Di, N) * 0.05, rn(Di)
    chunk = hidden[:, :Di * (Kw + N)]
    y_ref, conv_ref, ssm_ref = K.mamba_step_ref(chunk[:, :Di * Kw].reshape(B, Di, Kw), chunk[:, Di * Kw:].reshape(B, Di, N), xz,
                                                conv_w[:, 0], conv_b, xproj_w, dt_w, dt_b, A_log, D)
    dev = lambda t: t.cuda()
    y, new = ops.mamba_step(dev(hidden)[:, :Di * (Kw + N)], dev(xz), dev(conv_w), dev(conv_b), dev(xproj_w), dev(dt_w), dev(dt_b),
                            dev(A_log), dev(D), Kw, N)
    np.testing.assert_allclose(y.cpu(), y_ref, rtol=1e-4, atol=2e-5)
    np.testing.assert_array_equal(new[:, :Di * Kw].cpu(), conv_ref.reshape(B, -1))     # the window is a pure shift
    np.testing.assert_allclose(new[:, Di * Kw:].cpu(), ssm_ref.reshape(B, -1), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize('B,H,hd,S,steps', [(1, 8, 32, 64, 64), (3, 4, 64, 300, 290), (2, 2, 32, 1024, 520)])
def test_attn_decode_vs_oracle(ops, B, H, hd, S, steps):
    g = torch.Generator().manual_seed(H * hd)
    qkv = (torch.randn(steps, B, 3, H, hd, generator=g)).to(torch.bfloat16)
    slopes = K.alibi_slopes(H)
    cache = torch.zeros(B + 1, S, 2, H, hd, dtype=torch.bfloat16, device='cuda')       # cache batch may exceed the rows in use
    counter = torch.zeros(1, dtype=torch.int32, device='cuda')
    scale = hd ** -0.5
    check_at = sorted({0, 1, 63, 64, 255, 256, 257, steps - 1} & set(range(steps)))
    for t in range(steps):
        pos = counter if t % 2 else t                             # both position sources: device counter / host integer
        out = ops.attn_decode(qkv[t].cuda(), cache, pos, slopes.cuda(), scale)
        counter += 1
        if t in check_at:
            ref = K.attn_decode_ref(qkv[t, :, 0], qkv[:, :, 1].transpose(0, 1), qkv[:, :, 2].transpose(0, 1), t, slopes, scale)
            np.testing.assert_allclose(out.float().cpu(), ref, rtol=2e-2, atol=2e-2, err_msg=f'step {t}')
    # the cache now holds exactly the k, v that were fed
    np.testing.assert_array_equal(cache[:B, :steps, 0].float().cpu(), qkv[:, :, 1].transpose(0, 1).float())
    np.testing.assert_array_equal(cache[:B, :steps, 1].float().cpu(), qkv[:, :, 2].transpose(0, 1).float())
    assert float(cache[B].float().abs().sum()) == 0.0 and float(cache[:, steps:].float().abs().sum()) == 0.0


def test_attn_decode_full_cache_is_an_error(ops):
    qkv = torch.zeros(1, 3, 2, 32, dtype=torch.bfloat16, device='cuda')
    cache = torch.zeros(1, 4, 2, 2, 32, dtype=torch.bfloat16, device='cuda')
    with pytest.raises(RuntimeError):
        ops.attn_decode(qkv, cache, 4, None, 1.0)
    out = ops.attn_decode(qkv, cache, torch.full((1,), 4, dtype=torch.int32, device='cuda'), None, 1.0)
    assert torch.isnan(out.float()).all()                         # device counter: poisoned row instead of an out-of-bounds write


# ------------------------------------------------------------------------------------------------ layers vs the reference's steps
@pytest.mark.parametrize('lid', ROLLOUT_IDS)
def test_layer_steps_match_reference_recording(ops, lid):
    from offpolicy_rnn.models.rnn_base import RNNBase
    g = load_golden('rollout.npz')
    net = RNNBase(32, 32, [], ['linear'], [lid])
    pre = f'{lid}|p|'
    net.load_state_dict({k[len(pre):]: T(v) for k, v in g.items() if k.startswith(pre)})
    net.cuda()
    x = T(g['x']).cuda()
    hid = net.make_init_state(x.shape[0], x.device)
    hid[0] = T(g[f'{lid}|h0']).cuda()
    ys = []
    with torch.no_grad():
        for t in range(x.shape[1]):
            y, hid, _ = net.meta_forward(x[:, t:t + 1], hid)
            ys.append(y)
    np.testing.assert_allclose(torch.cat(ys, dim=1).cpu(), g[f'{lid}|y'], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(hid[0].cpu().reshape(-1), g[f'{lid}|hT'].reshape(-1), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize('lid,steps', [('cgpt_h8_l2_p0', 70), ('cgpt_h4_l1_p0_ml64_rms', 64)])
def test_cgpt_steps_match_packed_forward(ops, lid, steps):
    """Decoding token by token against the KV cache == the rows of the packed causal forward (the check the reference's
    own `main_onestep` prints, TransformerFlashAttention.py:124-138); bf16 attention on both sides."""
    from offpolicy_rnn.models.rnn_base import RNNBase
    torch.manual_seed(7)
    D, B = 32 * int(lid.split('_')[1][1:]), 2                      # head dim 32
    net = RNNBase(D, D, [], ['linear'], [lid]).cuda().eval()
    x = torch.randn(B, steps, D, device='cuda')
    with torch.no_grad():
        full, _, _ = net.meta_forward(x, net.make_init_state(B, x.device))
        hid = net.make_init_state(B, x.device)
        ys = []
        for t in range(steps):
            y, hid, _ = net.meta_forward(x[:, t:t + 1], hid)
            ys.append(y)
        assert hid[0].seqlen_offset == steps
        np.testing.assert_allclose(torch.cat(ys, dim=1).cpu(), full.cpu(), rtol=3e-2, atol=3e-2)
        if 'ml64' in lid:                                         # the cache is full now
            with pytest.raises(RuntimeError):
                net.meta_forward(x[:, :1], hid)


# ------------------------------------------------------------------------------------------------ hipGraph replay of the policy step
@pytest.mark.parametrize('rnn,algo', [('gru', 'sac'), ('smamba_s8_c4_b2_nln', 'sac'), ('gilr', 'td3'), ('lru', 'sac'),
                                      ('gilr_lstm', 'sac'), ('conv1d_3', 'sac'), ('mamba_s8_c3', 'td3'),
                                       ('cgpt_h1_l2_p0_ml32', 'td3')])
def test_graphed_policy_step_matches_eager(ops, rnn, algo):
    from offpolicy_rnn import alg_init
    from offpolicy_rnn.hip.graph_step import GraphedPolicyStep
    from offpolicy_rnn.utility.sample_utility import n2t_2dim
    from test_host_logic import make_parameter
    alg = alg_init(make_parameter(rnn, algo=algo, cuda_inference=True))
    dev = alg.device
    assert dev.type == 'cuda' and alg.graph_step is not None
    rs = np.random.RandomState(3)
    o, a, n = alg.obs_dim, alg.act_dim, 12
    obs, acts, rew = rs.randn(n + 1, 1, o), np.tanh(rs.randn(n + 1, 1, a)), rs.randn(n + 1, 1, 1)
    step = GraphedPolicyStep(alg.policy, dev)
    h0 = alg.policy.make_rnd_init_state(1, dev)
    step.load_hidden(h0)
    hid = h0
    for ep in range(2):                                            # second pass: a new episode through the SAME graph
        if ep == 1:
            hid = alg.policy.make_init_state(1, dev)
            step.load_hidden(hid)
        for t in range(n):
            with torch.no_grad():
                mean, _, _, _, hid, _ = alg.policy.forward(state=n2t_2dim(obs[t + 1], dev), lst_state=n2t_2dim(obs[t], dev),
                                                           lst_action=n2t_2dim(acts[t], dev), rnn_memory=hid,
                                                           reward=n2t_2dim(rew[t], dev))
            gmean, gsample, glogp = step(obs[t + 1], obs[t], acts[t], rew[t])
            tol = 3e-2 if rnn.startswith('cgpt') else 1e-5
            np.testing.assert_allclose(gmean, mean.reshape(1, -1).cpu().numpy(), rtol=tol, atol=tol, err_msg=f'{rnn} ep {ep} step {t}')
            assert np.isfinite(gsample).all() and np.isfinite(glogp).all()
            if algo == 'sac':
                assert np.abs(gsample).max() <= 1.0
    # parameters are read through their storage: an in-place change is seen by the next replay
    with torch.no_grad():
        for p in alg.policy.parameters():
            p.mul_(0.5)
        before = gmean
        after = step(obs[0], obs[1], acts[0], rew[0])[0]
    assert not np.allclose(before, after)


@pytest.mark.parametrize('name', ['gru_sac', 'gru_td3', 'gilr_sac', 'lru_sac', 'smamba_sac'])
def test_policy_step_vs_oracle(ops, name):
    """Whole policy step (encoders -> recurrent stack -> MLP head -> squashed Gaussian) on the reference's trained-run
    weights: graphed GPU step vs the CPU oracle's `policy_step`, same hidden carry, means and log-probs at fp32 tolerance."""
    import json
    import os
    from conftest import GOLDEN, nested
    from offpolicy_rnn import alg_init
    from oracle import network as NW
    from oracle.trainer import OracleTrainer, default_parameter
    from test_host_logic import make_parameter
    m = json.load(open(os.path.join(GOLDEN, 'train_meta.json')))[name]
    g = load_golden(f'train_{name}.npz')
    alg = alg_init(make_parameter(m['rnn'], algo=m['algo'], cuda_inference=True))
    alg.policy.load_state_dict(nested(g, 'policy0|'))
    par = default_parameter(rnn=m['rnn'], D=32, algo=m['algo'], policy_embedding_dim=16, value_embedding_dim=16,
                            policy_uni_model_input_mapping_dim=16, value_uni_model_input_mapping_dim=16)
    tr = OracleTrainer(par, 5, 3, 12, policy_state=nested(g, 'policy0|'), value_state=nested(g, 'value0|'))
    rs = np.random.RandomState(5)
    n = 10
    obs, acts, rew = rs.randn(n + 1, 1, 5), np.tanh(rs.randn(n + 1, 1, 3)), rs.randn(n + 1, 1, 1)
    h0 = alg.policy.make_rnd_init_state(1, alg.device)
    alg.graph_step.load_hidden(h0)
    hidden = [h[0].cpu().clone() for h in h0._data]
    for t in range(n):
        gmean, gsample, glogp = alg.graph_step(obs[t + 1], obs[t], acts[t], rew[t])
        with torch.no_grad():
            mean, sample, logp, hidden = NW.policy_step(tr.policy, tr.pcfg, T(obs[t + 1]), T(obs[t]), T(acts[t]), hidden, T(rew[t]),
                                                        noise=torch.zeros(1, 3), algo=m['algo'])
        np.testing.assert_allclose(gmean, mean, rtol=1e-4, atol=2e-5, err_msg=f'{name} step {t}')
    for got, want in zip(alg.graph_step._hidden._data, hidden):
        np.testing.assert_allclose(got[0].cpu(), want, rtol=1e-4, atol=2e-5)


def test_rollout_loop_uses_the_graph(ops):
    """The trainer's environment loop (sac.py) goes through the graphed step when sampling on the GPU."""
    from offpolicy_rnn import alg_init
    from test_host_logic import make_parameter
    alg = alg_init(make_parameter('smamba_s8_c4_b1_nln', cuda_inference=True))
    alg.env_reset()
    acts = [alg.sample_action() for _ in range(3)]
    assert alg.graph_step._graph is not None
    assert all(a.shape == (1, alg.act_dim) and np.isfinite(a).all() for a in acts)


@pytest.mark.parametrize('rnn,algo', [('smamba_s8_c4_b1_nln', 'sac'), ('cgpt_h1_l1_p0_ml32', 'td3'), ('gru', 'sac')])
def test_graphed_policy_step_with_several_environments(ops, rnn, algo):
    """The step kernels take B rows: one graph replay advances B independent environments (rows must not mix)."""
    from offpolicy_rnn import alg_init
    from offpolicy_rnn.hip.graph_step import GraphedPolicyStep
    from test_host_logic import make_parameter
    alg = alg_init(make_parameter(rnn, algo=algo, cuda_inference=True))
    B, n = 3, 6
    rs = np.random.RandomState(1)
    o, a = alg.obs_dim, alg.act_dim
    obs, acts, rew = rs.randn(n + 1, B, o), np.tanh(rs.randn(n + 1, B, a)), rs.randn(n + 1, B, 1)
    batched = GraphedPolicyStep(alg.policy, alg.device, batch_size=B)
    batched.load_hidden(None)
    singles = [GraphedPolicyStep(alg.policy, alg.device, batch_size=1) for _ in range(B)]
    for s1 in singles:
        s1.load_hidden(None)
    tol = 3e-2 if rnn.startswith('cgpt') else 1e-5
    for t in range(n):
        mean_b = batched(obs[t + 1], obs[t], acts[t], rew[t])[0]
        for r, s1 in enumerate(singles):
            mean_1 = s1(obs[t + 1, r:r + 1], obs[t, r:r + 1], acts[t, r:r + 1], rew[t, r:r + 1])[0]
            np.testing.assert_allclose(mean_b[r:r + 1], mean_1, rtol=tol, atol=tol, err_msg=f'{rnn} step {t} row {r}')
