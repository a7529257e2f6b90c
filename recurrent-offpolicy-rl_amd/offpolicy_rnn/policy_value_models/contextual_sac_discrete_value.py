"""Ensemble critic for discrete action spaces (reference
offpolicy_rnn/policy_value_models/contextual_sac_discrete_value.py:9-116): context embedding + phi_s(state) -> `efc-<E>`
MLP -> Q for every action [E, rows, T', A].  The `action` argument of forward() is accepted and ignored, as upstream."""
from typing import Optional, Tuple

import torch

from ..models.RNNHidden import RNNHidden
from ..models.contextual_model import ContextualModel
from ..models.rnn_base import ACTIVATIONS
from . import _inputs
from .utils import nearest_power_of_two, nearest_power_of_two_half
from ..models.linear import Linear


class ContextualSACDiscreteValue(ContextualModel):
    def __init__(self, state_dim, action_dim, embedding_size, embedding_hidden, embedding_activations, embedding_layer_type,
                 uni_model_hidden, uni_model_activations, uni_model_layer_type, fix_rnn_length, uni_model_input_mapping_dim: int = 0,
                 reward_input=False, last_action_input=True, last_state_input=False, separate_encoder=False):
        self.embedding_state_dim = state_dim
        if embedding_size == 'auto':
            embedding_size = nearest_power_of_two_half(state_dim)
        if uni_model_input_mapping_dim == 'auto':
            uni_model_input_mapping_dim = nearest_power_of_two(state_dim + action_dim)
        cum_dim = _inputs.build_encoders(self, state_dim, action_dim, reward_input, last_action_input, last_state_input, separate_encoder)
        uni_in = state_dim
        self.state_input_encoder = torch.nn.Identity()
        if uni_model_input_mapping_dim > 0 and separate_encoder:
            self.state_input_encoder = Linear(state_dim, uni_model_input_mapping_dim)
            uni_in = uni_model_input_mapping_dim
            uni_model_input_mapping_dim = 0
        super().__init__(embedding_input_size=cum_dim, embedding_size=embedding_size, embedding_hidden=embedding_hidden,
                         embedding_activations=embedding_activations, embedding_layer_type=embedding_layer_type,
                         uni_model_input_size=uni_in, uni_model_output_size=action_dim, uni_model_hidden=uni_model_hidden,
                         uni_model_activations=uni_model_activations, uni_model_layer_type=uni_model_layer_type,
                         fix_rnn_length=fix_rnn_length, uni_model_input_mapping_dim=uni_model_input_mapping_dim,
                         uni_model_input_mapping_activation=embedding_activations[-1], name='ContextualSACValue')
        self.uni_model_input_mapping_activation_func = ACTIVATIONS[embedding_activations[-1]]()
        _inputs.register_encoders(self)
        if separate_encoder:
            self.contextual_register_rnn_base_module(self.state_input_encoder, 'state_input_encoder_q')
        self.state_dim, self.action_dim = state_dim, action_dim
        self.finalize_parameters()

    def get_embedding_input(self, state, lst_state, lst_action, reward) -> torch.Tensor:
        return _inputs.embedding_input(self, state, lst_state, lst_action, reward)

    def state_encoding(self, state):
        s = self.state_input_encoder(state)
        return self.uni_model_input_mapping_activation_func(s) if self.separate_encoder else s

    def forward(self, state, lst_state, lst_action, action, rnn_memory: Optional[RNNHidden], reward, detach_embedding=False
                ) -> Tuple[torch.Tensor, torch.Tensor, RNNHidden, Optional[RNNHidden]]:
        emb_in = self.get_embedding_input(state, lst_state, lst_action, reward)
        value, rnn_memory, emb, full = self.meta_forward(emb_in, self.state_encoding(state), rnn_memory, detach_embedding)
        return value, emb, rnn_memory, full

    def forward_embedding(self, state, lst_state, lst_action, rnn_memory, reward):
        return self.get_embedding(self.get_embedding_input(state, lst_state, lst_action, reward), rnn_memory)
