"""CPU restatement of the full-trajectory recurrent SAC / TD3 (REDQ, separate RNN learning rate) update.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows
  offpolicy_rnn/algorithm/sac_full_length_rnn_ensembleQ.py:297-467   train_one_batch
  offpolicy_rnn/algorithm/sac_full_length_rnn_redq.py:16-49          REDQ target / actor loss
  offpolicy_rnn/algorithm/td3_full_length_rnn_redq.py:14-51          TD3 twin
  offpolicy_rnn/algorithm/sac_full_length_rnn_redq_sep_optim.py:37-92 AdamW parameter groups
  offpolicy_rnn/utility/q_value_guard.py:4-45                        target clamp
It is also the reported CPU baseline of bench.py (kind "port"): `gru_impl='aten'` routes the GRU layer
through ATen's fused CPU GRU exactly like the reference's torch.nn.GRU does.
"""
import math
import time
from types import SimpleNamespace
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from . import network as NW
from .buffer import OracleBuffer, Transition


def default_parameter(rnn='gru', D=256, algo='sac', **over) -> SimpleNamespace:
    """Published architecture (gen_tmuxp_mamba_pomdp.py:43-86) with the RNN layer id swapped;
    remaining fields are the defaults of offpolicy_rnn/parameter/ParameterSAC.py:18-306."""
    p = SimpleNamespace(
        alg_name=('sac' if algo == 'sac' else 'td3') + '_rnn_full_horizon_redQ_sep_optim', base_algorithm=algo,
        seed=1, policy_lr=3e-4, rnn_policy_lr=1e-5, policy_l2_norm=0.0, policy_update_per=1,
        policy_max_gradnorm=None, alpha_lr=1e-4, value_lr=1e-3, rnn_value_lr=1e-4, value_max_gradnorm=None,
        value_l2_norm=0.0, reward_input=False, last_state_input=True, no_last_action_input=False,
        state_action_encoder=True, randomize_mask=False, valid_number_post_randomized=256,
        policy_uni_model_input_mapping_dim=128, value_uni_model_input_mapping_dim=128,
        no_alpha_auto_tune=False, sac_alpha=0.2,
        value_hidden_size=[D, D], value_activations=['elu', 'elu', 'linear'], value_layer_type=['efc-8'] * 3,
        value_embedding_hidden_size=[D, D], value_embedding_activations=['elu', 'elu', 'linear'],
        value_embedding_layer_type=['fc', rnn, 'fc'], value_embedding_dim=128,
        policy_hidden_size=[D, D], policy_activations=['elu', 'elu', 'linear'], policy_layer_type=['fc'] * 3,
        policy_embedding_hidden_size=[D, D], policy_embedding_activations=['elu', 'elu', 'linear'],
        policy_embedding_layer_type=['fc', rnn, 'fc'], policy_embedding_dim=128,
        utd=1, policy_utd=1, redq_m=2, gamma=0.99, sac_tau=0.995, target_entropy_ratio=1.5,
        max_buffer_transition_num=int(1e6), sac_batch_size=1024, sample_std=0.1,
        target_action_noise_std=0.04, target_action_noise_clip=0.12,
    )
    for k, v in over.items():
        setattr(p, k, v)
    return p


def model_cfg(par, obs_dim, act_dim, which: str) -> dict:
    """SAC._make_policy_args / _make_value_args (offpolicy_rnn/algorithm/sac.py:199-239)."""
    g = lambda n: getattr(par, f'{which}_{n}')
    return dict(state_dim=obs_dim, action_dim=act_dim, embedding_size=g('embedding_dim'),
                embedding_hidden=g('embedding_hidden_size'), embedding_activations=g('embedding_activations'),
                embedding_layer_type=g('embedding_layer_type'), uni_model_hidden=g('hidden_size'),
                uni_model_activations=g('activations'), uni_model_layer_type=g('layer_type'),
                reward_input=par.reward_input, last_action_input=not par.no_last_action_input,
                last_state_input=par.last_state_input, uni_model_input_mapping_dim=g('uni_model_input_mapping_dim'),
                separate_encoder=par.state_action_encoder, algo=par.base_algorithm)


class QValueGuard:
    """offpolicy_rnn/utility/q_value_guard.py:4-45."""
    def __init__(self, decay_ratio=1 - 1e-3):
        self._min, self._max, self._init, self._decay = 1000000, -1000000, True, decay_ratio

    def clamp(self, v):
        if self._init:
            self._min, self._max, self._init = v.min().item(), v.max().item(), False
        return v.clamp(min=self._min, max=self._max)

    def update(self, v):
        vmin, vmax = v.min().item(), v.max().item()
        self._min, self._max = min(self._min, vmin), max(self._max, vmax)
        if self._decay < 1:
            self._min = self._decay * self._min + (1 - self._decay) * vmin
            self._max = self._decay * self._max + (1 - self._decay) * vmax


def skip_len(par) -> int:
    """_get_skip_len (sac_full_length_rnn_ensembleQ.py:57-68)."""
    s = 0
    for lt in (par.value_layer_type + par.value_embedding_layer_type + par.policy_layer_type + par.policy_embedding_layer_type):
        if 'mamba' in lt or 'conv1d' in lt:                       # smamba, mamba (s6) and conv1d carry a conv window
            s = max(s, NW.parse_layer_id(lt)['d_conv'])
    return s + 1


def allow_nest_stack(par) -> bool:
    """SAC.allow_nest_stack_trajs (sac.py:130-138)."""
    for lt in (par.value_layer_type + par.value_embedding_layer_type + par.policy_layer_type + par.policy_embedding_layer_type):
        if 'transformer' in lt or 'gru' in lt:
            return False
    return True


class OracleTrainer:
    def __init__(self, par, obs_dim, act_dim, max_traj_len, smamba_semantics='gpu', gru_impl='ref',
                 policy_state=None, value_state=None, discrete=False):
        self.par, self.obs_dim, self.act_dim = par, obs_dim, act_dim
        self.algo = par.base_algorithm
        self.discrete = discrete                                   # act_dim discrete actions (sac.py:49,72-79)
        if discrete:
            par.no_alpha_auto_tune = True                          # sac.py:74
        self.fw = dict(smamba_semantics=smamba_semantics, gru_impl=gru_impl)
        self.pcfg = dict(model_cfg(par, obs_dim, act_dim, 'policy'), discrete=discrete)
        self.vcfg = dict(model_cfg(par, obs_dim, act_dim, 'value'), discrete=discrete)
        clone = lambda sd: {m: {k: v.clone().float() for k, v in d.items()} for m, d in sd.items()}
        self.policy = clone(policy_state) if policy_state is not None else NW.init_model(self.pcfg, 'policy')
        self.value = clone(value_state) if value_state is not None else NW.init_model(self.vcfg, 'value')
        self.target_value = clone(self.value)                      # sac.py:69 hard update
        for net in (self.policy, self.value):
            for t in NW.flat_params(net):
                t.requires_grad_(True)
        a0 = math.log(par.sac_alpha) if par.no_alpha_auto_tune else 0.0   # sac.py:75-78
        self.log_alpha = torch.tensor([a0], dtype=torch.float32, requires_grad=True)
        if self.algo == 'td3':
            # td3_full_length_rnn_ensembleQ.py:21-22 flips the flag only AFTER SAC.__init__ built log_alpha,
            # so a TD3 trainer starts (and stays) at log_alpha = 0 unless the user passed --no_alpha_auto_tune
            par.no_alpha_auto_tune = True
        self.target_entropy = par.target_entropy_ratio if discrete else -float(act_dim) * par.target_entropy_ratio  # sac.py:79
        self.opt_policy = torch.optim.AdamW(self._groups(self.policy, par.rnn_policy_lr, par.policy_l2_norm),
                                            lr=par.policy_lr, weight_decay=par.policy_l2_norm)
        self.opt_value = torch.optim.AdamW(self._groups(self.value, par.rnn_value_lr, par.value_l2_norm),
                                           lr=par.value_lr, weight_decay=par.value_l2_norm)
        self.opt_alpha = torch.optim.AdamW([self.log_alpha], lr=par.alpha_lr)   # sac.py:90 (default weight_decay!)
        self.buffer = OracleBuffer(par.max_buffer_transition_num, max_traj_len, additional_history_len=skip_len(par))
        self.guard = QValueGuard(1.0 if discrete else 1 - 1e-3)  # sac_full_length_rnn_ensembleQ.py:43-46
        self.nest = allow_nest_stack(par)
        self.grad_num = 0

    @staticmethod
    def _groups(net, rnn_lr, wd):
        """prepare_param_list (sac_full_length_rnn_redq_sep_optim.py:49-66): the whole embedding_model at rnn_lr."""
        groups = []
        for k, mod in net.items():
            if k == 'embedding_model':
                groups.append({'params': list(mod.values()), 'lr': rnn_lr, 'weight_decay': wd})
            else:
                groups.append({'params': list(mod.values())})
        return groups

    # ----------------------------------------------------------------------------------------------
    def fill_synthetic(self, n_traj, T, seed=0):
        """Synthetic Gaussian trajectories (SURVEY.md section 8(d) 'Synthetic inputs')."""
        rs = np.random.RandomState(seed)
        for _ in range(n_traj):
            obs = rs.randn(T + 1, self.obs_dim)
            act = np.tanh(rs.randn(T, self.act_dim))
            rew = rs.randn(T)
            for t in range(T):
                self.buffer.mem_push(Transition(
                    state=obs[t:t + 1], last_state=obs[t - 1:t] if t > 0 else np.zeros((1, self.obs_dim)),
                    last_action=act[t - 1:t] if t > 0 else np.zeros((1, self.act_dim)), action=act[t:t + 1],
                    next_state=obs[t + 1:t + 2], reward=float(rew[t]), logp=None, mask=1, start=(t == 0),
                    done=(t == T - 1), reward_input=np.array([[rew[t - 1] if t > 0 else 0.0]]), timeout=(t == T - 1)))

    # ----------------------------------------------------------------------------------------------
    def train_one_batch(self) -> Dict:
        par = self.par
        policy_losses = {}
        policy_update_cnt = 0
        for utd_idx in range(par.utd):
            batch, batch_size, valid, table = self.buffer.sample_trajs(
                par.sac_batch_size, randomize_mask=par.randomize_mask,
                valid_number_post_randomized=par.valid_number_post_randomized,
                equalize_data_of_each_traj=True, nest_stack_trajs=self.nest)
            n2t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(torch.float32)
            (state, last_state, action, last_action, next_state, done, mask, reward, reward_input, timeout,
             rnn_start) = [n2t(getattr(batch, k)) for k in ('state', 'last_state', 'action', 'last_action', 'next_state',
                                                             'done', 'mask', 'reward', 'reward_input', 'timeout', 'start')]
            valid = n2t(valid)
            # flag surgery (sac_full_length_rnn_ensembleQ.py:338-342)
            total_start, total_valid = rnn_start.clone(), valid.clone()
            total_valid[torch.where(torch.diff(valid, dim=-2) == 1)] = 1
            total_start[torch.where(torch.diff(total_start, dim=-2) == -1)] = 0
            done[timeout > 0] = 0
            alpha = self.log_alpha.exp().detach()
            # per-row sequence tables (:358-366)
            am = torch.from_numpy(table).to(torch.float32)
            am = torch.cat((am, torch.zeros(am.shape[0], state.shape[-2] - am.shape[1])), dim=-1)
            tam = torch.cat((am[..., 1:], torch.zeros(am.shape[0], 1)), dim=-1).to(torch.int)
            am = am.to(torch.int)
            f_target = NW.Flags(total_start, total_valid, tam)
            f_online = NW.Flags(rnn_start, valid, am)

            target_Q = self._target_Q(state, action, next_state, done, reward, f_target, alpha)
            self.guard.update(target_Q * mask)                      # :387
            valid_num = mask.sum()
            # critic step (:392 -> :261-295)
            q, _ = NW.value_forward(self.value, self.vcfg, state, last_state, last_action, action, f_online,
                                    reward_input, **self.fw)
            if self.discrete:                                       # select_with_action (:158)
                q = q.gather(-1, action.long().unsqueeze(0).expand(q.shape[0], -1, -1, -1))
            Q_loss = ((q - target_Q.unsqueeze(0)).pow(2).sum(dim=0) * mask).sum() / valid_num
            self.opt_value.zero_grad()
            Q_loss.backward()
            q_grad_norm = 0
            if par.value_max_gradnorm is not None:
                q_grad_norm = torch.nn.utils.clip_grad_norm_(NW.flat_params(self.value), par.value_max_gradnorm).item()
            self.opt_value.step()
            # soft update (:395 -> rnn_base.py:490-491)
            with torch.no_grad():
                for m in self.value:
                    for k in self.value[m]:
                        tp = self.target_value[m][k]
                        tp.copy_(tp * par.sac_tau + (1 - par.sac_tau) * self.value[m][k])
            # actor + alpha step (:405-432)
            if self.grad_num % par.policy_update_per == 0 and (utd_idx + 1) / par.utd * par.policy_utd > policy_update_cnt:
                mean, _, sample, logp = NW.policy_forward(self.policy, self.pcfg, state, last_state, last_action,
                                                          f_online, reward_input, algo=self.algo,
                                                          sample_std=par.sample_std, **self.fw)
                act_in = sample if self.algo == 'sac' else mean     # td3_full_length_rnn_redq.py:45
                qpi, _ = NW.value_forward(self.value, self.vcfg, state, last_state, last_action, act_in, f_online,
                                          reward_input, detach_embedding=True, **self.fw)
                qbar = qpi.mean(dim=0)                              # sac_full_length_rnn_redq.py:46
                if self.discrete:                                   # sac_full_length_rnn_redq.py:84-86
                    actor_loss = (((((alpha * logp) - qbar) * logp.exp()).sum(dim=-1, keepdim=True)) * mask).sum() / valid_num
                    logp = (logp * logp.exp()).sum(dim=-1, keepdim=True)        # logged form (:425)
                elif self.algo == 'sac':
                    actor_loss = (((alpha * logp) - qbar) * mask).sum() / valid_num
                else:
                    actor_loss = ((-qbar) * mask).sum() / valid_num
                self.opt_policy.zero_grad()
                actor_loss.backward()
                pi_grad_norm = 0
                if par.policy_max_gradnorm is not None:
                    pi_grad_norm = torch.nn.utils.clip_grad_norm_(NW.flat_params(self.policy), par.policy_max_gradnorm).item()
                self.opt_policy.step()
                if not par.no_alpha_auto_tune:
                    alpha_loss = -((self.log_alpha * (logp + self.target_entropy).detach()) * mask).sum() / valid_num
                    self.opt_alpha.zero_grad()
                    alpha_loss.backward()
                    self.opt_alpha.step()
                    with torch.no_grad():
                        self.log_alpha.clamp_max_(1)                # :422
                    policy_losses['alpha_loss'] = alpha_loss.item()
                policy_losses['log_prob'] = ((logp * mask).sum() / valid_num).item()
                policy_update_cnt += 1
                policy_losses['actor_loss'] = (actor_loss.item(),)  # a 1-tuple in the reference (:430)
                policy_losses['policy_grad_norm'] = pi_grad_norm
                policy_losses['policy_l2_norm_square'] = NW.l2_norm_square(self.policy).item()
        return {
            'critic_loss': Q_loss.item(), 'log_alpha': self.log_alpha.item(), 'real_batch_size': batch_size,
            'real_batch_traj_num': state.shape[0], 'target_q_max': target_Q.abs().max().item(),
            'value_grad_norm': q_grad_norm, 'clip_min': self.guard._min, 'clip_max': self.guard._max,
            'q1_l2_norm_square': NW.l2_norm_square(self.value).item(),
            'average_traj_len': self.buffer.size / len(self.buffer), 'amp_scalar_pi': 0, 'amp_scalar_q': 0,
            **policy_losses,
        }

    def _target_Q(self, state, action, next_state, done, reward, flags, alpha):
        par = self.par
        if self.discrete:                                           # sac_full_length_rnn_redq.py:52-72
            with torch.no_grad():
                onehot = F.one_hot(action.squeeze(-1).long(), num_classes=self.act_dim).float()
                _, _, sample, logp = NW.policy_forward(self.policy, self.pcfg, next_state, state, onehot, flags, reward, **self.fw)
                nq, _ = NW.value_forward(self.target_value, self.vcfg, next_state, state, onehot, sample, flags, reward, **self.fw)
                idx = np.random.permutation(nq.shape[0])[:par.redq_m]
                v = ((nq[idx, :].min(dim=0).values - alpha * logp) * logp.exp()).sum(dim=-1, keepdim=True)
                return reward + (1 - done) * par.gamma * self.guard.clamp(v)
        with torch.no_grad():
            mean, _, sample, logp = NW.policy_forward(self.policy, self.pcfg, next_state, state, action, flags, reward,
                                                      algo=self.algo, sample_std=par.sample_std, **self.fw)
            if self.algo == 'td3':                                  # td3_full_length_rnn_redq.py:21-25
                noise = torch.clamp(torch.randn_like(mean) * par.target_action_noise_std,
                                    -par.target_action_noise_clip, par.target_action_noise_clip)
                sample = torch.clamp(mean + noise, -1, 1)
            nq, _ = NW.value_forward(self.target_value, self.vcfg, next_state, state, action, sample, flags, reward, **self.fw)
            idx = np.random.permutation(nq.shape[0])[:par.redq_m]  # sac_full_length_rnn_redq.py:28
            mn = nq[idx, :].min(dim=0).values
            if self.algo == 'sac':
                mn = mn - alpha * logp
            return reward + (1 - done) * par.gamma * self.guard.clamp(mn)


def time_cpu_baseline(rnn='gru', B=64, T=1024, obs=17, act=6, updates=3, warmup=1, threads=None, seed=0, algo='sac'):
    """bench.py `cpu_baseline` leg: the GRU SAC trainer restatement on the host cores.
    Returns dict(value=env-steps/s trained, seconds_per_update, cores, sample)."""
    if threads:
        torch.set_num_threads(threads)
    torch.manual_seed(seed)
    np.random.seed(seed)
    par = default_parameter(rnn=rnn, sac_batch_size=B * T - 1, algo=algo)
    tr = OracleTrainer(par, obs, act, T, gru_impl='aten')
    tr.fill_synthetic(2 * B, T, seed)
    for _ in range(warmup):
        tr.train_one_batch()
    t0 = time.time()
    n = 0
    for _ in range(updates):
        n += tr.train_one_batch()['real_batch_size']
        tr.grad_num += 1
    dt = time.time() - t0
    return dict(value=n / dt, seconds_per_update=dt / updates, cores=torch.get_num_threads(),
                sample=f'{updates} update(s) of {rnn} {algo.upper()}-REDQ at B={B},T={T},D=256 after {warmup} warm-up update(s)')


def time_cpu_rollout(rnn='gru', obs=17, act=6, steps=200, warmup=10, threads=1, seed=0):
    """bench.py --mode rollout `cpu_baseline` leg: the policy's one-token step (the reference samples on the CPU unless
    --cuda_inference, algorithm/sac.py:48-50) on the host cores.  Returns dict(value=policy steps/s, us_per_step, ...)."""
    torch.set_num_threads(threads)
    torch.manual_seed(seed)
    par = default_parameter(rnn=rnn)
    tr = OracleTrainer(par, obs, act, 8)
    spec = tr.pcfg['embedding_layer_type']
    hidden = [torch.zeros(1, NW.hidden_size_of(l, 256, 256)) for l in spec if NW.is_rnn(l)]
    g = torch.Generator().manual_seed(seed)
    s, ls, la, r = torch.randn(1, obs, generator=g), torch.randn(1, obs, generator=g), torch.randn(1, act, generator=g), torch.randn(1, 1, generator=g)
    with torch.no_grad():
        for i in range(warmup + steps):
            if i == warmup:
                t0 = time.time()
            _, sample, _, hidden = NW.policy_step(tr.policy, tr.pcfg, s, ls, la, hidden, r)
    dt = time.time() - t0
    return dict(value=steps / dt, us_per_step=1e6 * dt / steps, cores=torch.get_num_threads(),
                sample=f'{steps} policy steps of {rnn} (D=256, obs={obs}, act={act}) on {threads} thread(s) after {warmup} warm-up')
