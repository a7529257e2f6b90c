"""Tanh-Gaussian actor (reference offpolicy_rnn/policy_value_models/contextual_sac_policy_single_head.py:11-129).
The head arithmetic (clamp, sample, tanh-squash log-prob and its backward) is one fused HIP kernel (`ops.tanh_gaussian`)."""
from typing import Optional, Tuple

import torch

from ..hip import ops
from ..models.RNNHidden import RNNHidden
from ..models.contextual_model import ContextualModel
from . import _inputs
from ..utility import rng
from .utils import nearest_power_of_two, nearest_power_of_two_half


class ContextualSACPolicySingleHead(ContextualModel):
    MAX_LOG_STD = 2.0
    MIN_LOG_STD = -20.0

    def __init__(self, state_dim, action_dim, embedding_size, embedding_hidden, embedding_activations, embedding_layer_type,
                 uni_model_hidden, uni_model_activations, uni_model_layer_type, fix_rnn_length, uni_model_input_mapping_dim: int = 0,
                 reward_input=False, last_action_input=True, last_state_input=False, separate_encoder=False, output_logstd=True,
                 name='ContextualSACPolicy'):
        if uni_model_activations[-1] != 'linear':
            uni_model_activations = list(uni_model_activations[:-1]) + ['linear']
        if uni_model_layer_type[-1] != 'fc':
            raise NotImplementedError(f'It is not supported to construct {uni_model_layer_type[-1]} logstd and mean head!')
        if embedding_size == 'auto':
            embedding_size = nearest_power_of_two_half(state_dim)
        if uni_model_input_mapping_dim == 'auto':
            uni_model_input_mapping_dim = nearest_power_of_two(state_dim)
        cum_dim = _inputs.build_encoders(self, state_dim, action_dim, reward_input, last_action_input, last_state_input, separate_encoder)
        super().__init__(embedding_input_size=cum_dim, embedding_size=embedding_size, embedding_hidden=embedding_hidden,
                         embedding_activations=embedding_activations, embedding_layer_type=embedding_layer_type,
                         uni_model_input_size=state_dim, uni_model_output_size=action_dim * 2 if output_logstd else action_dim,
                         uni_model_hidden=uni_model_hidden, uni_model_activations=uni_model_activations,
                         uni_model_layer_type=uni_model_layer_type, fix_rnn_length=fix_rnn_length,
                         uni_model_input_mapping_dim=uni_model_input_mapping_dim,
                         uni_model_input_mapping_activation=embedding_activations[-1], name=name)
        _inputs.register_encoders(self)
        self.state_dim, self.action_dim = state_dim, action_dim
        self.finalize_parameters()

    def get_embedding_input(self, state, lst_state, lst_action, reward) -> torch.Tensor:
        return _inputs.embedding_input(self, state, lst_state, lst_action, reward)

    def forward(self, state, lst_state, lst_action, rnn_memory: Optional[RNNHidden], reward=None, detach_embedding=False
                ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, RNNHidden, Optional[RNNHidden]]:
        emb_in = self.get_embedding_input(state, lst_state, lst_action, reward)
        out, rnn_memory, emb, full = self.meta_forward(emb_in, state, rnn_memory, detach_embedding)
        action_mean, action_sample, log_prob = self.process_model_out(out)
        return action_mean, emb, action_sample, log_prob, rnn_memory, full

    def process_model_out(self, out2, noise=None):
        """out2 = (logstd | mean).  noise defaults to a fresh N(0, I) draw (torch generator of out2's device)."""
        if noise is None:
            noise = rng.randn(out2.shape[:-1] + (out2.shape[-1] // 2,), out2.device, out2.dtype)
        return ops.tanh_gaussian(out2, noise)

    def forward_embedding(self, state, lst_state, lst_action, rnn_memory, reward):
        return self.get_embedding(self.get_embedding_input(state, lst_state, lst_action, reward), rnn_memory)
