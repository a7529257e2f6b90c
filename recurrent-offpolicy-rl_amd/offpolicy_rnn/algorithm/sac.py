"""SAC base trainer: construction (env, seeds, actor / critics, optimizers, replay buffer), the environment loop and
checkpoint I/O (reference offpolicy_rnn/algorithm/sac.py:33-420).  The per-update work lives in the full-trajectory
sub-classes (`train_one_batch`).  Out of scope here, as in SURVEY.md section 8: the evaluation worker pool
(`eval_inprocess`, sac.py:285-300,364-379) - `train()` logs training returns only."""
import math
import os
import random
import time
from typing import Dict, List, Union

import numpy as np
import torch

from .._compat import Logger, smart_logger
from ..buffers.transition_buffer.replay_memory import MemoryArray, Transition
from ..env_utils.make_env import make_env
from ..policy_value_models.make_models import make_policy_model, make_value_model
from ..utility.count_parameters import count_parameters
from ..utility.sample_utility import n2t_2dim, norm_act, t2n, unorm_act
from ..utility.timer import Timer
from .flat_adamw import FlatAdamW


class SAC:
    def __init__(self, parameter):
        self.parameter = parameter
        self.timer = Timer()
        self.logger = Logger(log_name=self.parameter.short_name)
        self.parameter.set_config_path(os.path.join(self.logger.output_dir, 'config'))
        self.parameter.save_config()
        self.logger(self.parameter)
        # environment
        self.env_name = parameter.env_name
        self.env_info = make_env(self.env_name, parameter.seed)
        self.eval_env_info = make_env(self.env_name, parameter.seed + 1)
        self.env, self.eval_env = self.env_info['train_env'], self.eval_env_info['train_env']
        self.max_episode_steps = self.env_info['max_trajectory_len']
        self.discrete_env = not self.env_info['act_continuous']
        self.obs_dim = self.env.observation_space.shape[0]
        self.act_dim = self.env_info['act_dim']
        self._seed(parameter.seed)
        # networks
        self.policy_args = self._make_policy_args(parameter)
        self.value_args = self._make_value_args(parameter)
        self.device = self._pick_device()
        # The reference samples on the CPU unless --cuda_inference (sac.py:45-49).  This build has no CPU forward (every layer
        # is a HIP kernel), so on a GPU the policy always lives - and samples - on the device: the flag is implied.
        if self.device.type == 'cuda' and not parameter.cuda_inference:
            self.logger('cuda_inference is implied on a GPU: the policy stays resident on the device (no CPU forward in this build)')
        self.sample_device = self.device
        self.base_algorithm = getattr(parameter, 'base_algorithm', 'sac')
        self.policy = make_policy_model(self.policy_args, self.base_algorithm, self.discrete_env)
        self.values = [make_value_model(self.value_args, self.base_algorithm, self.discrete_env) for _ in range(parameter.value_net_num)]
        self.target_values = [make_value_model(self.value_args, self.base_algorithm, self.discrete_env) for _ in range(parameter.value_net_num)]
        for v in self.values + self.target_values:
            v.to(self.device)
        self.policy.to(self.sample_device)
        self._value_update(tau=0.0)
        if self.discrete_env:
            parameter.no_alpha_auto_tune = True                  # reference sac.py:72-74: fixed temperature for discrete actions
        if getattr(parameter, 'no_alpha_auto_tune', False):
            alpha0 = math.log(parameter.sac_alpha)
        else:
            alpha0 = 0.0
        self.log_sac_alpha = torch.tensor([alpha0], dtype=torch.float32, device=self.device).requires_grad_(True)
        self.target_entropy = parameter.target_entropy_ratio if self.discrete_env else \
            -float(np.prod(self.env.action_space.shape)) * parameter.target_entropy_ratio
        # optimizers: one flat AdamW per network (a single learning rate here; the *_sep_optim trainers regroup)
        self.optimizer_policy = FlatAdamW(self.policy.store, lambda m: parameter.policy_lr, lambda m: parameter.policy_l2_norm)
        self.optimizers_value = [FlatAdamW(v.store, lambda m: parameter.value_lr, lambda m: parameter.value_l2_norm) for v in self.values]
        self.optimizer_alpha = torch.optim.AdamW([self.log_sac_alpha], lr=parameter.alpha_lr)   # default weight decay, as upstream
        for v in self.values:
            v.train()
        for v in self.target_values:
            v.eval()
        self.policy.train()
        self.replay_buffer = MemoryArray(parameter.max_buffer_transition_num, smart_logger.get_customized_value('MAX_TRAJ_STEP'))
        # rollout state
        self.state_np = np.zeros((1, self.obs_dim))
        self.last_action_np = np.zeros((1, self.act_dim))
        self.last_state_np = np.zeros((1, self.obs_dim))
        self.reward_np = np.zeros((1, 1))
        self.sample_hidden = None
        # on the GPU the per-step policy forward is replayed as one hipGraph (hip/graph_step.py); RESEL_GRAPH_ROLLOUT=0: eager
        self.graph_step = None
        if self.sample_device.type == 'cuda' and os.environ.get('RESEL_GRAPH_ROLLOUT', '1') != '0' and not self.discrete_env:
            from ..hip.graph_step import GraphedPolicyStep
            self.graph_step = GraphedPolicyStep(self.policy, self.sample_device, batch_size=1)
        self.sample_num = 0
        self.grad_num = 0
        self.start_time = time.time()
        self.allow_nest_stack = self.allow_nest_stack_trajs()
        self.logger(f'policy parameter num: {count_parameters(self.policy)}; value[0] parameter num: {count_parameters(self.values[0])}')

    @staticmethod
    def _pick_device():
        if torch.cuda.is_available():
            return torch.device('cuda', torch.cuda.current_device())
        return torch.device('cpu')

    @property
    def optimizer_value(self):
        return self.optimizers_value[0]

    def allow_nest_stack_trajs(self):
        """Rows may pack several trajectories unless a layer cannot honour resets (gru ignores rnn_start)."""
        for net in (self.values[0].uni_network, self.values[0].embedding_network, self.policy.uni_network, self.policy.embedding_network):
            for lid in net.layer_type:
                if 'transformer' in lid or 'gru' in lid:
                    return False
        return True

    def _seed(self, seed: int):
        np.random.seed(seed)
        random.seed(seed + 1)
        torch.manual_seed(seed + 2)
        if torch.cuda.is_available():
            torch.cuda.manual_seed(seed + 3)
            torch.cuda.manual_seed_all(seed + 4)
        self.env.seed(seed + 5)
        self.env.action_space.seed(seed + 6)
        self.env.observation_space.seed(seed + 7)

    def _value_update(self, tau):
        """target <- tau * target + (1 - tau) * online (tau = 0: hard copy)."""
        for value, target in zip(self.values, self.target_values):
            target.copy_weight_from(value, tau)

    def _common_args(self, p, which: str) -> Dict[str, Union[int, float, List[int]]]:
        g = lambda k: getattr(p, f'{which}_{k}')
        return {
            'state_dim': self.obs_dim, 'action_dim': self.act_dim, 'embedding_size': g('embedding_dim'),
            'embedding_hidden': g('embedding_hidden_size'), 'embedding_activations': g('embedding_activations'),
            'embedding_layer_type': g('embedding_layer_type'), 'uni_model_hidden': g('hidden_size'),
            'uni_model_activations': g('activations'), 'uni_model_layer_type': g('layer_type'), 'fix_rnn_length': p.rnn_fix_length,
            'reward_input': p.reward_input, 'last_action_input': not p.no_last_action_input,
            'last_state_input': bool(getattr(p, 'last_state_input', False)),
            'uni_model_input_mapping_dim': g('uni_model_input_mapping_dim'),
            'separate_encoder': bool(getattr(p, 'state_action_encoder', False)),
        }

    def _make_policy_args(self, parameter):
        args = self._common_args(parameter, 'policy')
        if getattr(parameter, 'base_algorithm', 'sac') == 'td3':
            args['sample_std'] = parameter.sample_std
        return args

    def _make_value_args(self, parameter):
        return self._common_args(parameter, 'value')

    # ------------------------------------------------------------------------------------------ environment loop
    def _init_sample_hidden(self):
        # (the reference's branches are swapped - randomize_first_hidden picks the ZERO state, sac.py:143-148; kept)
        if self.parameter.randomize_first_hidden:
            return self.policy.make_init_state(1, self.sample_device)
        return self.policy.make_rnd_init_state(1, self.sample_device)

    def env_reset(self):
        self.state_np = np.asarray(self.env.reset()).reshape((1, -1))
        self.last_action_np = np.zeros((1, self.act_dim))
        self.last_state_np = np.zeros((1, self.obs_dim))
        self.reward_np = np.zeros((1, 1))
        self.sample_hidden = self._init_sample_hidden()
        if self.graph_step is not None:
            self.graph_step.load_hidden(self.sample_hidden)

    def sample_action(self) -> np.ndarray:
        """One policy step on the current rollout state (reference sac.py:319-326) -> sampled action [1, act_dim]."""
        if self.graph_step is not None:
            return self.graph_step(self.state_np, self.last_state_np, self.last_action_np, self.reward_np)[1].reshape(1, -1)
        with torch.no_grad():
            _, _, act_sample, _, self.sample_hidden, _ = self.policy.forward(
                state=n2t_2dim(self.state_np, self.sample_device), lst_state=n2t_2dim(self.last_state_np, self.sample_device),
                lst_action=n2t_2dim(self.last_action_np, self.sample_device), rnn_memory=self.sample_hidden,
                reward=n2t_2dim(self.reward_np, self.sample_device))
        return t2n(act_sample).reshape(1, -1)

    def env_step(self, next_obs, act, reward, done):
        self.last_state_np = self.state_np.copy()
        self.state_np = np.asarray(next_obs).copy().reshape((1, -1))
        if self.discrete_env:                                     # one-hot of the action index (reference :167-169)
            self.last_action_np = np.zeros((1, self.act_dim))
            self.last_action_np[..., int(np.asarray(act).reshape(-1)[0])] = 1
        else:
            self.last_action_np = np.asarray(act).copy().reshape((1, -1))
        self.reward_np = np.array([[reward]])
        if done:
            self.env_reset()

    def _push(self, act_normalized, next_state, reward, done, traj_len):
        self.replay_buffer.mem_push(Transition(
            state=self.state_np, last_state=self.last_state_np, last_action=self.last_action_np, action=act_normalized,
            next_state=np.asarray(next_state).reshape((1, -1)), reward=reward, logp=None, mask=1, done=done,
            timeout=traj_len >= self.max_episode_steps, start=traj_len == 1, reward_input=self.reward_np))

    def warmup(self):
        self.env_reset()
        n = 0
        while n < self.parameter.random_num:
            done, traj_len = False, 0
            while not done:
                act = self.env.action_space.sample()
                next_state, reward, done, _ = self.env.step(act)
                act_n = norm_act(act, self.env.action_space)
                traj_len += 1
                self._push(np.asarray(act_n).reshape((1, -1)), next_state, reward, done, traj_len)
                self.env_step(next_state, act_n, reward, done)
                n += 1
        return n

    def train_one_batch(self) -> Dict:
        return {}

    def train(self):
        self.sample_num += self.warmup()
        self.env_reset()
        update = self.train_one_batch
        graphed = None
        if os.environ.get('RESEL_GRAPH_UPDATE', '1') != '0':  # default: the whole update as one hipGraph replay per recurring batch shape (graphed_update.py)
            from .graphed_update import GraphedUpdate
            why = GraphedUpdate.refusal(self)
            if why:
                self.logger(f'updates are launched eagerly ({why})')
            else:
                graphed = GraphedUpdate(self)
                update = graphed.step
        try:
            self._train_loop(update)
        finally:
            if graphed is not None:             # detach the process-wide dropout base: later eager trainers of this process draw from torch's generator again
                graphed.close()

    def _train_loop(self, update):
        ep_ret, ep_len = 0.0, 0
        for it in range(self.parameter.total_iteration):
            self.policy.train()
            self.policy.to(self.sample_device)
            for _ in range(self.parameter.step_per_iteration):
                act_sample = self.sample_action()
                act_env = unorm_act(act_sample[0], self.env.action_space)
                next_state, reward, done, _ = self.env.step(int(act_env) if self.discrete_env else act_env)
                ep_ret += reward
                ep_len += 1
                self._push(act_sample, next_state, reward, done, ep_len)
                self.env_step(next_state, act_sample, reward, done)
                if done:
                    self.logger.add_tabular_data(tb_prefix='Train', EpRet=ep_ret, EpLength=ep_len)
                    ep_ret, ep_len = 0.0, 0
                if self.sample_num % self.parameter.update_interval == 0 and self.sample_num >= self.parameter.start_train_num:
                    self.logger.add_tabular_data(tb_prefix='train', **update())
                    self.grad_num += 1
                self.sample_num += 1
            self.logger.log_tabular('iteration', it, tb_prefix='timestep')
            self.logger.log_tabular('timestep', self.sample_num, tb_prefix='timestep')
            self.logger.log_tabular('grad_num', self.grad_num, tb_prefix='timestep')
            self.logger.log_tabular('time', time.time() - self.start_time, tb_prefix='timestep')
            self.logger.dump_tabular()
            if it % 25 == 0:
                self.save()

    # ------------------------------------------------------------------------------------------ checkpoints
    def save(self, model_dir=None):
        path = os.path.join(self.logger.output_dir, 'model') if model_dir is None else model_dir
        self.policy.save(path)
        for i in range(len(self.values)):
            self.values[i].save(path, index=f'{i}')
            self.target_values[i].save(path, index=f'{i}-target')
        torch.save(self.log_sac_alpha, os.path.join(path, 'log_sac_alpha.pt'))

    def load(self, model_dir=None, load_policy=True, load_value=True):
        path = os.path.join(self.logger.output_dir, 'model') if model_dir is None else model_dir
        if load_policy:
            self.policy.load(path, map_location=self.sample_device)
            if self.graph_step is not None:
                self.graph_step.invalidate()
        if load_value:
            for i in range(len(self.values)):
                self.values[i].load(path, index=f'{i}', map_location=self.device)
                self.target_values[i].load(path, index=f'{i}-target', map_location=self.device)
        # in place: optimizer_alpha holds this tensor (rebinding it, as the reference does, silently freezes the temperature)
        with torch.no_grad():
            self.log_sac_alpha.copy_(torch.load(os.path.join(path, 'log_sac_alpha.pt'), map_location=self.device))
