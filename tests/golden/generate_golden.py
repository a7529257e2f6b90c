#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by IMPORTING THE REFERENCE (build container only).

    python tests/golden/generate_golden.py            # needs /root/reference, writes tests/golden/*.npz|*.json

The reference (FanmingL/Recurrent-Offpolicy-RL, pure Python) never travels to the GPU box; only the
vectors written here do.  The harness follows SURVEY.md appendix A: `offpolicy_rnn` is registered as a
namespace package (its __init__ pulls gym + the env zoo), and `smart_logger`, `gym`,
`offpolicy_rnn.env_utils.make_env`, `selective_scan_cuda` are replaced by inert stubs.
Fixtures are data only (inputs, parameters, expected outputs) - no reference source text.
"""
import json
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get('RESEL_REFERENCE', '/root/reference')
OUT = os.path.dirname(os.path.abspath(__file__))


# ------------------------------------------------------------------------------------------------ stubs
def install_stubs(obs_dim=5, act_dim=3, T=12):
    pkg = types.ModuleType('offpolicy_rnn')
    pkg.__path__ = [os.path.join(REF, 'offpolicy_rnn')]
    sys.modules['offpolicy_rnn'] = pkg

    sl = types.ModuleType('smart_logger')

    class Logger:
        def __init__(self, log_name=None, **kw):
            self.output_dir = '/tmp/resel_golden_log'

        def __call__(self, *a, **k):
            pass

        def add_tabular_data(self, *a, **k):
            pass

        log_tabular = dump_tabular = sync_log_to_remote = add_tabular_data

    sl.Logger = Logger
    sl.init_config = lambda *a, **k: None
    sl.get_customized_value = lambda name: 1000
    sl.experiment_config = types.SimpleNamespace(EXPERIMENT_TARGET='golden')
    par_pkg = types.ModuleType('smart_logger.parameter')
    par_mod = types.ModuleType('smart_logger.parameter.ParameterTemplate')

    class ParameterTemplate:
        short_name = 'golden'

        def __init__(self, config_path=None, debug=False):
            pass

        def set_config_path(self, p):
            pass

        def save_config(self):
            pass

    par_mod.ParameterTemplate = ParameterTemplate
    sys.modules.update({'smart_logger': sl, 'smart_logger.parameter': par_pkg,
                        'smart_logger.parameter.ParameterTemplate': par_mod})

    gym = types.ModuleType('gym')

    class Space:
        pass

    class Box(Space):
        def __init__(self, low, high, shape):
            self.low = np.full(shape, low, dtype=np.float32)
            self.high = np.full(shape, high, dtype=np.float32)
            self.shape = shape

        def seed(self, s):
            pass

        def sample(self):
            return np.random.uniform(-1, 1, self.shape)

    class Discrete(Space):
        def __init__(self, n):
            self.n, self.shape = n, ()

        def seed(self, s):
            pass

        def sample(self):
            return int(np.random.randint(self.n))

    class Env:
        pass

    gym.Space, gym.Env = Space, Env
    gym.spaces = types.SimpleNamespace(Box=Box, Discrete=Discrete)
    sys.modules['gym'] = gym
    sys.modules['gym.spaces'] = gym.spaces

    class FakeEnv(Env):
        def __init__(self):
            self.observation_space = Box(-np.inf, np.inf, (ENV['obs'],))
            self.action_space = Discrete(ENV['act']) if ENV.get('discrete') else Box(-1, 1, (ENV['act'],))
            self.t = 0

        def seed(self, s):
            pass

        def reset(self):
            self.t = 0
            return np.random.randn(ENV['obs'])

        def step(self, a):
            self.t += 1
            return np.random.randn(ENV['obs']), float(np.random.randn()), self.t >= ENV['T'], {}

    eu = types.ModuleType('offpolicy_rnn.env_utils')
    eu.__path__ = []
    me = types.ModuleType('offpolicy_rnn.env_utils.make_env')

    def make_env(name, seed):
        return dict(train_env=FakeEnv(), eval_env=FakeEnv(), train_tasks=[], eval_tasks=[None], max_rollouts_per_task=1,
                    max_trajectory_len=ENV['T'], obs_dim=ENV['obs'], act_dim=ENV['act'], act_continuous=not ENV.get('discrete'),
                    seed=seed, multiagent=False)

    me.make_env = make_env
    sys.modules['offpolicy_rnn.env_utils'] = eu
    sys.modules['offpolicy_rnn.env_utils.make_env'] = me
    sys.modules['selective_scan_cuda'] = types.ModuleType('selective_scan_cuda')
    sys.argv = ['golden']


ENV = dict(obs=5, act=3, T=12)


def t2n(t):
    return t.detach().cpu().numpy().copy()      # copy: parameters are updated in place later


def flat_sd(model_sd, prefix=''):
    """{module: {key: tensor}} -> {'module|key': ndarray}"""
    return {f'{prefix}{m}|{k}': t2n(v) for m, d in model_sd.items() for k, v in d.items()}


# ------------------------------------------------------------------------------------------------ fixtures
def synth_traj(rs, T, obs, act):
    o = rs.randn(T + 1, obs)
    a = np.tanh(rs.randn(T, act))
    r = rs.randn(T)
    return o, a, r


def push_traj(buffer, Transition, o, a, r, early_done=False):
    T = len(a)
    obs, act = o.shape[1], a.shape[1]
    for t in range(T):
        buffer.mem_push(Transition(
            state=o[t:t + 1], last_state=o[t - 1:t] if t > 0 else np.zeros((1, obs)),
            last_action=a[t - 1:t] if t > 0 else np.zeros((1, act)), action=a[t:t + 1], next_state=o[t + 1:t + 2],
            reward=float(r[t]), logp=None, mask=1, start=(t == 0), done=(t == T - 1),
            reward_input=np.array([[r[t - 1] if t > 0 else 0.0]]), timeout=(t == T - 1) and not early_done))


def gen_sample_trajs():
    from offpolicy_rnn.buffers.transition_buffer.nested_replay_memory import NestedMemoryArray
    from offpolicy_rnn.buffers.transition_buffer.replay_memory import Transition
    out = {}
    cases = [dict(name='skip2_nonest', hist=1, nest=False, rmask=False, lens=[7, 7, 7, 7], bs=13),
             dict(name='skip18_nest', hist=17, nest=True, rmask=False, lens=[5, 9, 3, 12, 6, 4], bs=20),
             dict(name='skip2_nest_rmask', hist=1, nest=True, rmask=True, lens=[6, 11, 4, 8, 10], bs=18),
             dict(name='skip2_fixedT', hist=1, nest=False, rmask=False, lens=[12] * 6, bs=12 * 3 - 1)]
    for c in cases:
        buf = NestedMemoryArray(1000, 12, additional_history_len=c['hist'])
        rs = np.random.RandomState(7)
        for L in c['lens']:
            o, a, r = synth_traj(rs, L, 4, 2)
            push_traj(buf, Transition, o, a, r, early_done=(L != 12))
        np.random.seed(123)
        res, total, valid, table = buf.sample_trajs(c['bs'], None, randomize_mask=c['rmask'],
                                                    valid_number_post_randomized=9, equalize_data_of_each_traj=True,
                                                    random_trunc_traj=False, nest_stack_trajs=c['nest'])
        n = c['name']
        out[f'{n}|cfg'] = np.array([c['hist'], int(c['nest']), int(c['rmask']), c['bs'], 12] + c['lens'], dtype=np.int64)
        for f in res._fields:
            v = getattr(res, f)
            if v is not None:
                out[f'{n}|{f}'] = np.array(v, copy=True)
        out[f'{n}|total'] = np.array(total)
        out[f'{n}|valid'] = valid.copy()
        out[f'{n}|table'] = table.copy()
    np.savez_compressed(os.path.join(OUT, 'sample_trajs.npz'), **out)
    print('sample_trajs:', len(out), 'arrays')


def gen_selective_scan():
    from offpolicy_rnn.models.smamba.mamba_ssm.ops.selective_scan_interface_new import selective_scan_ref
    from offpolicy_rnn.models.s6.selective_scan.cpu_scan import selective_scan_cpu  # noqa: F401 (second oracle, survey 8(c))
    out = {}
    for name, (B, Di, L, N) in dict(n16=(2, 8, 24, 16), n32=(2, 16, 37, 32), n64=(1, 8, 19, 64)).items():
        g = torch.Generator().manual_seed(11)
        r = lambda *s: torch.randn(*s, generator=g)
        u, delta, z = r(B, Di, L), r(B, Di, L) * 0.5, r(B, Di, L)
        A = -torch.exp(r(Di, N) * 0.3)
        Bm, Cm = r(B, N, L), r(B, N, L)
        D, db = r(Di), r(Di) * 0.1
        start = torch.zeros(B, 1, L)
        start[:, :, 0] = 1
        start[0, :, L // 2] = 1
        start[-1, :, L - 3] = 1
        ins = [u, delta, A, Bm, Cm, D, z, db]
        for t in ins:
            t.requires_grad_(True)
        o, last = selective_scan_ref(u, delta, A, Bm, Cm, start.expand(B, Di, L), D, z, db, True, True)
        w = r(B, Di, L)
        (o * w).sum().backward()
        for k, t in zip('u delta A Bm Cm D z delta_bias'.split(), ins):
            out[f'{name}|{k}'] = t2n(t)
            out[f'{name}|d{k}'] = t2n(t.grad)
        out[f'{name}|start'] = t2n(start[:, 0])
        out[f'{name}|out'] = t2n(o)
        out[f'{name}|last_state'] = t2n(last)
        out[f'{name}|dout'] = t2n(w)
    np.savez_compressed(os.path.join(OUT, 'selective_scan.npz'), **out)
    print('selective_scan ok')


def gen_layers():
    """Sequence layers through the reference's own modules (CPU paths)."""
    from offpolicy_rnn.models.rnn_base import RNNBase
    import offpolicy_rnn.models.smamba.mamba as mamba_mod
    from offpolicy_rnn.models.smamba.mamba_ssm.ops.selective_scan_interface_new import selective_scan_ref
    out = {}
    B, L, D = 2, 21, 32
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(B, L, D, generator=g)
    start = torch.zeros(B, L, 1)
    start[:, 0] = 1
    start[0, 9] = 1
    start[1, 15:] = 1
    mask = torch.ones(B, L, 1)
    mask[1, 15:] = 0
    mask[0, 8] = 0
    w = torch.randn(B, L, D, generator=g)
    for lid in ['gru', 'gilr', 'lru', 'smamba_s8_c6_b2_nln', 'smamba_s16_c4_b1', 'smamba_s8_c5_b1_ff',
                'gilr_lstm', 'conv1d_5', 'mamba_s8_c3', 'mamba_s4_c5_noff']:          # appended: earlier entries keep their draws
        torch.manual_seed(3)
        net = RNNBase(D, D, [], ['linear'], [lid])
        with torch.no_grad():          # the zero-initialised biases would hide bias-handling bugs
            for n_, p_ in net.named_parameters():
                if 'bias' in n_ and p_.abs().sum() == 0:
                    p_.copy_(torch.randn(p_.shape, generator=g) * 0.1)
        hid = net.make_init_state(B, torch.device('cpu'))
        hid.set_rnn_start(start)
        hid.set_mask(mask)
        x = x0.clone().requires_grad_(True)
        tag = lid
        if lid.startswith('smamba'):
            # (a) GPU-path semantics on CPU: forward_sequential + selective_scan_ref
            orig_fwd, orig_fn = mamba_mod.Mamba.forward, mamba_mod.selective_scan_fn
            mamba_mod.selective_scan_fn = selective_scan_ref
            mamba_mod.Mamba.forward = lambda self, xx, hidden=None, rnn_start=None, mask=None: (
                self.forward_sequential(xx, mask, rnn_start), hidden)
            y, _, _ = net.meta_forward(x, hid)
            (y * w).sum().backward()
            out[f'{tag}|y_seq'] = t2n(y)
            out[f'{tag}|dx_seq'] = t2n(x.grad)
            for n_, p_ in net.named_parameters():
                if p_.grad is not None:
                    out[f'{tag}|gseq|{n_}'] = t2n(p_.grad)
            mamba_mod.Mamba.forward, mamba_mod.selective_scan_fn = orig_fwd, orig_fn
            net.zero_grad()
            x = x0.clone().requires_grad_(True)
            # (b) what the reference does with CPU tensors: the per-step loop (ignores start / mask)
            y, _, _ = net.meta_forward(x, net.make_init_state(B, torch.device('cpu')))
            out[f'{tag}|y_step'] = t2n(y)
        else:
            y, _, _ = net.meta_forward(x, hid)
            (y * w).sum().backward()
            out[f'{tag}|y'] = t2n(y)
            out[f'{tag}|dx'] = t2n(x.grad)
            for n_, p_ in net.named_parameters():
                if p_.grad is not None:       # e.g. GILRLayer.layer_norm is constructed but never used
                    out[f'{tag}|g|{n_}'] = t2n(p_.grad)
        for n_, p_ in net.state_dict().items():
            out[f'{tag}|p|{n_}'] = t2n(p_)
    out['x'], out['start'], out['mask'], out['w'] = t2n(x0), t2n(start), t2n(mask), t2n(w)
    np.savez_compressed(os.path.join(OUT, 'layers.npz'), **out)
    print('layers ok')


def gen_rollout():
    """T = 1 rollout steps through the reference's own modules: one token per meta_forward call, hidden carried
    (what algorithm/sac.py:319-326 does between updates).  Per-step outputs + the final hidden state."""
    from offpolicy_rnn.models.rnn_base import RNNBase
    out = {}
    B, L, D = 2, 9, 32
    g = torch.Generator().manual_seed(11)
    x0 = torch.randn(B, L, D, generator=g)
    for lid in ['gru', 'gilr', 'lru', 'smamba_s8_c6_b2_nln', 'smamba_s16_c4_b1', 'smamba_s8_c5_b1_ff',
                'gilr_lstm', 'conv1d_5', 'mamba_s8_c3', 'mamba_s4_c5_noff']:
        torch.manual_seed(4)
        net = RNNBase(D, D, [], ['linear'], [lid])
        with torch.no_grad():
            for n_, p_ in net.named_parameters():
                if 'bias' in n_ and p_.abs().sum() == 0:
                    p_.copy_(torch.randn(p_.shape, generator=g) * 0.1)
        hid = net.make_rnd_init_state(B, torch.device('cpu'))
        out[f'{lid}|h0'] = t2n(hid[0])
        ys = []
        with torch.no_grad():
            for t in range(L):
                y, hid, _ = net.meta_forward(x0[:, t:t + 1], hid)
                ys.append(y)
        out[f'{lid}|y'] = t2n(torch.cat(ys, dim=1))
        out[f'{lid}|hT'] = t2n(hid[0])
        for n_, p_ in net.state_dict().items():
            out[f'{lid}|p|{n_}'] = t2n(p_)
    out['x'] = t2n(x0)
    np.savez_compressed(os.path.join(OUT, 'rollout.npz'), **out)
    print('rollout ok')


def make_parameter(rnn, D=32, algo='sac', **over):
    from offpolicy_rnn.parameter.ParameterSAC import Parameter
    p = Parameter()
    p.parse()
    p.alg_name = ('sac' if algo == 'sac' else 'td3') + '_rnn_full_horizon_redQ_sep_optim'
    p.base_algorithm = algo
    p.value_net_num = 1
    p.test_nprocess = 1
    for w in ('value', 'policy'):
        setattr(p, f'{w}_embedding_layer_type', ['fc', rnn, 'fc'])
        setattr(p, f'{w}_embedding_activations', ['elu', 'elu', 'linear'])
        setattr(p, f'{w}_embedding_hidden_size', [D, D])
        setattr(p, f'{w}_hidden_size', [D, D])
        setattr(p, f'{w}_activations', ['elu', 'elu', 'linear'])
        setattr(p, f'{w}_embedding_dim', 16)
        setattr(p, f'{w}_uni_model_input_mapping_dim', 16)
    p.value_layer_type = ['efc-8'] * 3
    p.policy_layer_type = ['fc'] * 3
    p.state_action_encoder = True
    p.last_state_input = True
    p.alpha_lr = 1e-4
    p.policy_update_per = 1
    p.max_buffer_transition_num = 5000
    for k, v in over.items():
        setattr(p, k, v)
    return p


def gen_models_and_train():
    from offpolicy_rnn.algorithm.sac_full_length_rnn_redq_sep_optim import SACFullLengthRNNREDQ_SEP_OPTIM
    from offpolicy_rnn.algorithm.td3_full_length_rnn_redq_sep_optim import TD3FullLengthRNNREDQ_SEP_OPTIM
    from offpolicy_rnn.buffers.transition_buffer.replay_memory import Transition
    obs, act, T = ENV['obs'], ENV['act'], ENV['T']
    meta = {}
    for name, rnn, algo, lens in [('gru_sac', 'gru', 'sac', [T] * 6), ('gru_td3', 'gru', 'td3', [T] * 6),
                                  ('gilr_sac', 'gilr', 'sac', [T, 5, 7, T, 4, 9, 6]),
                                  ('lru_sac', 'lru', 'sac', [T, 5, 7, T, 4, 9, 6]),
                                  ('smamba_sac', 'smamba_s8_c3_b2_nln', 'sac', [T, 5, 7, T, 4, 9, 6])]:
        torch.manual_seed(100)
        np.random.seed(100)
        par = make_parameter(rnn, algo=algo, sac_batch_size=int(sum(lens) * 0.6))
        cls = SACFullLengthRNNREDQ_SEP_OPTIM if algo == 'sac' else TD3FullLengthRNNREDQ_SEP_OPTIM
        alg = cls(par)
        out = {}
        out.update(flat_sd(alg.policy.state_dict(), 'policy0|'))
        out.update(flat_sd(alg.values[0].state_dict(), 'value0|'))
        rs = np.random.RandomState(9)
        for L in lens:
            o, a, r = synth_traj(rs, L, obs, act)
            push_traj(alg.replay_buffer, Transition, o, a, r, early_done=(L != T))
        # --- one policy / value forward with pinned noise (F6) -------------------------------------
        np.random.seed(5)
        batch, _, valid, table = alg.replay_buffer.sample_trajs(par.sac_batch_size, None, equalize_data_of_each_traj=True,
                                                                nest_stack_trajs=alg.allow_nest_stack)
        f32 = lambda a_: torch.from_numpy(np.array(a_, copy=True)).float()
        st, ls, la, ac, rs_, ri = map(f32, (batch.state, batch.last_state, batch.last_action, batch.action, batch.start,
                                            batch.reward_input))
        hp = alg.policy.make_init_state(st.shape[0], torch.device('cpu'))
        hp.set_rnn_start(rs_)
        hp.set_mask(f32(valid))
        torch.manual_seed(77)
        mean, emb, samp, logp, _, _ = alg.policy.forward(st, ls, la, hp, ri)
        hv = alg.values[0].make_init_state(st.shape[0], torch.device('cpu'))
        hv.set_rnn_start(rs_)
        hv.set_mask(f32(valid))
        q, qemb, _, _ = alg.values[0].forward(st, ls, la, ac, hv, ri)
        torch.manual_seed(77)
        noise = torch.randn_like(mean)
        for k, v in dict(fw_state=st, fw_last_state=ls, fw_last_action=la, fw_action=ac, fw_start=rs_, fw_valid=f32(valid),
                         fw_noise=noise, fw_mean=mean, fw_emb=emb, fw_sample=samp, fw_logp=logp, fw_q=q, fw_qemb=qemb).items():
            out[k] = t2n(v)
        # --- three consecutive updates (F8) -----------------------------------------------------------
        torch.manual_seed(200)
        np.random.seed(200)
        logs = []
        for _ in range(3):
            log = alg.train_one_batch()
            alg.grad_num += 1
            log = {k: (float(v[0]) if isinstance(v, tuple) else float(v)) for k, v in log.items()}
            logs.append(log)
        out.update(flat_sd(alg.policy.state_dict(), 'policy3|'))
        out.update(flat_sd(alg.values[0].state_dict(), 'value3|'))
        out.update(flat_sd(alg.target_values[0].state_dict(), 'target3|'))
        out['log_alpha3'] = t2n(alg.log_sac_alpha)
        np.savez_compressed(os.path.join(OUT, f'train_{name}.npz'), **out)
        meta[name] = dict(rnn=rnn, algo=algo, lens=lens, sac_batch_size=par.sac_batch_size, logs=logs,
                          skip_len=alg._get_skip_len(), nest=bool(alg.allow_nest_stack))
        alg.process_pool.shutdown()
        print('train', name, logs[-1]['critic_loss'])
    with open(os.path.join(OUT, 'train_meta.json'), 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)


def gen_discrete_train():
    """Discrete-action SAC-REDQ (categorical actor, all-action critic; reference contextual_sac_discrete_*.py and
    sac_full_length_rnn_redq.py:52-89): initial weights, one forward of actor / critic, three consecutive updates."""
    from offpolicy_rnn.algorithm.sac_full_length_rnn_redq_sep_optim import SACFullLengthRNNREDQ_SEP_OPTIM
    from offpolicy_rnn.buffers.transition_buffer.replay_memory import Transition
    obs, act, T = ENV['obs'], 4, ENV['T']
    ENV['discrete'], keep_act = True, ENV['act']
    ENV['act'] = act
    meta = {}
    for name, rnn, lens in [('gru_sac_discrete', 'gru', [T] * 6), ('gilr_sac_discrete', 'gilr', [T, 5, 7, T, 4, 9, 6])]:
        torch.manual_seed(100)
        np.random.seed(100)
        par = make_parameter(rnn, algo='sac', sac_batch_size=int(sum(lens) * 0.6), sac_alpha=0.2)
        alg = SACFullLengthRNNREDQ_SEP_OPTIM(par)
        assert alg.discrete_env
        out = {}
        out.update(flat_sd(alg.policy.state_dict(), 'policy0|'))
        out.update(flat_sd(alg.values[0].state_dict(), 'value0|'))
        rs = np.random.RandomState(9)
        for L in lens:
            o, r = rs.randn(L + 1, obs), rs.randn(L)
            a = rs.randint(act, size=(L, 1)).astype(np.float64)
            oh = np.eye(act)[a[:, 0].astype(int)]
            for t in range(L):
                alg.replay_buffer.mem_push(Transition(
                    state=o[t:t + 1], last_state=o[t - 1:t] if t > 0 else np.zeros((1, obs)),
                    last_action=oh[t - 1:t] if t > 0 else np.zeros((1, act)), action=a[t:t + 1], next_state=o[t + 1:t + 2],
                    reward=float(r[t]), logp=None, mask=1, start=(t == 0), done=(t == L - 1),
                    reward_input=np.array([[r[t - 1] if t > 0 else 0.0]]), timeout=(t == L - 1) and L == T))
        np.random.seed(5)
        batch, _, valid, table = alg.replay_buffer.sample_trajs(par.sac_batch_size, None, equalize_data_of_each_traj=True,
                                                                nest_stack_trajs=alg.allow_nest_stack)
        f32 = lambda a_: torch.from_numpy(np.array(a_, copy=True)).float()
        st, ls, la, ac, rs_, ri = map(f32, (batch.state, batch.last_state, batch.last_action, batch.action, batch.start,
                                            batch.reward_input))
        hp = alg.policy.make_init_state(st.shape[0], torch.device('cpu'))
        hp.set_rnn_start(rs_)
        hp.set_mask(f32(valid))
        mean, emb, samp, logp, _, _ = alg.policy.forward(st, ls, la, hp, ri)
        hv = alg.values[0].make_init_state(st.shape[0], torch.device('cpu'))
        hv.set_rnn_start(rs_)
        hv.set_mask(f32(valid))
        q, qemb, _, _ = alg.values[0].forward(st, ls, la, ac, hv, ri)
        for k, v in dict(fw_state=st, fw_last_state=ls, fw_last_action=la, fw_action=ac, fw_start=rs_, fw_valid=f32(valid),
                         fw_mean=mean.float(), fw_emb=emb, fw_logp=logp, fw_q=q, fw_qemb=qemb).items():
            out[k] = t2n(v)
        torch.manual_seed(200)
        np.random.seed(200)
        logs = []
        for _ in range(3):
            log = alg.train_one_batch()
            alg.grad_num += 1
            logs.append({k: (float(v[0]) if isinstance(v, tuple) else float(v)) for k, v in log.items()})
        out.update(flat_sd(alg.policy.state_dict(), 'policy3|'))
        out.update(flat_sd(alg.values[0].state_dict(), 'value3|'))
        out.update(flat_sd(alg.target_values[0].state_dict(), 'target3|'))
        np.savez_compressed(os.path.join(OUT, f'train_{name}.npz'), **out)
        meta[name] = dict(rnn=rnn, algo='sac', lens=lens, sac_batch_size=par.sac_batch_size, logs=logs, n_actions=act,
                          sac_alpha=0.2, skip_len=alg._get_skip_len(), nest=bool(alg.allow_nest_stack))
        alg.process_pool.shutdown()
        print('train', name, logs[-1]['critic_loss'])
    ENV['discrete'], ENV['act'] = False, keep_act
    with open(os.path.join(OUT, 'train_discrete_meta.json'), 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)


def gen_checkpoints():
    """Checkpoint files exactly as the reference writes them (`SAC.save`, algorithm/sac.py; `<name>-<index>-<module>.pt`,
    models/contextual_model.py:135-143) for the initial networks of two of the trained-run fixtures: same seeds, so the
    tensors equal the `policy0|` / `value0|` entries of train_<name>.npz."""
    from offpolicy_rnn.algorithm.sac_full_length_rnn_redq_sep_optim import SACFullLengthRNNREDQ_SEP_OPTIM
    T = ENV['T']
    for name, rnn, lens in [('gru_sac', 'gru', [T] * 6), ('smamba_sac', 'smamba_s8_c3_b2_nln', [T, 5, 7, T, 4, 9, 6])]:
        torch.manual_seed(100)
        np.random.seed(100)
        par = make_parameter(rnn, algo='sac', sac_batch_size=int(sum(lens) * 0.6))
        alg = SACFullLengthRNNREDQ_SEP_OPTIM(par)
        path = os.path.join(OUT, f'ckpt_{name}')
        os.makedirs(path, exist_ok=True)
        alg.save(path)
        alg.process_pool.shutdown()
        print('checkpoint', name, sorted(os.listdir(path)))


def gen_layer_ids():
    """F9: layer-id string -> constructed hyper-parameters + hidden-state width."""
    from offpolicy_rnn.models.rnn_base import RNNBase
    table = {}
    for lid in ['gru', 'gilr', 'lru', 'smamba', 'smamba_s32_c16_b2_nln', 'smamba_b1_c8_s64_ff', 'smamba_s8_c3_b3',
                'gilr_lstm', 'conv1d', 'conv1d_7', 'mamba', 'mamba_s8_c6_noff']:
        net = RNNBase(32, 32, [], ['linear'], [lid])
        lay = net.layer_list[0]
        e = dict(hidden=int(net.rnn_hidden_state_input_size[0]), nparam=int(sum(p.numel() for p in net.parameters())))
        if lid.startswith('smamba'):
            e.update(d_conv=lay.d_conv, d_state=lay.layers[0].mixer.d_state, block_num=lay.block_num,
                     rms_norm=bool(lay.rms_norm), use_ff=bool(lay.use_ff))
        table[lid] = e
    with open(os.path.join(OUT, 'layer_ids.json'), 'w') as f:
        json.dump(table, f, indent=1, sort_keys=True)
    print('layer ids ok')


if __name__ == '__main__':
    assert os.path.isdir(REF), f'{REF} not found: fixtures can only be regenerated in the build container'
    install_stubs()
    torch.set_num_threads(4)
    gen_sample_trajs()
    gen_selective_scan()
    gen_layers()
    gen_rollout()
    gen_layer_ids()
    gen_models_and_train()
    gen_checkpoints()
    gen_discrete_train()
