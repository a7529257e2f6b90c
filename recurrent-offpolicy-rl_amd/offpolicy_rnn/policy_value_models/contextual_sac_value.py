"""Ensemble critic (reference offpolicy_rnn/policy_value_models/contextual_sac_value.py:9-126): context embedding +
[phi_s(state), phi_a(action)] -> `efc-<E>` MLP -> Q [E, rows, T', 1]."""
from typing import Optional, Tuple

import torch

from ..models.RNNHidden import RNNHidden
from ..models.contextual_model import ContextualModel
from ..models.rnn_base import ACTIVATIONS
from . import _inputs
from .utils import nearest_power_of_two, nearest_power_of_two_half
from ..models.linear import Linear


class ContextualSACValue(ContextualModel):
    def __init__(self, state_dim, action_dim, embedding_size, embedding_hidden, embedding_activations, embedding_layer_type,
                 uni_model_hidden, uni_model_activations, uni_model_layer_type, fix_rnn_length, uni_model_input_mapping_dim: int = 0,
                 reward_input=False, last_action_input=True, last_state_input=False, separate_encoder=False, name='ContextualSACValue'):
        self.embedding_state_dim = state_dim
        if embedding_size == 'auto':
            embedding_size = nearest_power_of_two_half(state_dim)
        if uni_model_input_mapping_dim == 'auto':
            uni_model_input_mapping_dim = nearest_power_of_two(state_dim + action_dim)
        cum_dim = _inputs.build_encoders(self, state_dim, action_dim, reward_input, last_action_input, last_state_input, separate_encoder)
        uni_in = state_dim + action_dim
        self.state_input_encoder = self.action_input_encoder = torch.nn.Identity()
        if uni_model_input_mapping_dim > 0 and separate_encoder:
            self.state_input_encoder = Linear(state_dim, uni_model_input_mapping_dim)
            self.action_input_encoder = Linear(action_dim, uni_model_input_mapping_dim)
            uni_in = uni_model_input_mapping_dim * 2
            uni_model_input_mapping_dim = 0         # the two encoders replace the generic input mapping network
        super().__init__(embedding_input_size=cum_dim, embedding_size=embedding_size, embedding_hidden=embedding_hidden,
                         embedding_activations=embedding_activations, embedding_layer_type=embedding_layer_type,
                         uni_model_input_size=uni_in, uni_model_output_size=1, uni_model_hidden=uni_model_hidden,
                         uni_model_activations=uni_model_activations, uni_model_layer_type=uni_model_layer_type,
                         fix_rnn_length=fix_rnn_length, uni_model_input_mapping_dim=uni_model_input_mapping_dim,
                         uni_model_input_mapping_activation=embedding_activations[-1], name=name)
        self.uni_model_input_mapping_activation_func = ACTIVATIONS[embedding_activations[-1]]()
        _inputs.register_encoders(self)
        if separate_encoder:
            self.contextual_register_rnn_base_module(self.state_input_encoder, 'state_input_encoder_q')
            self.contextual_register_rnn_base_module(self.action_input_encoder, 'action_input_encoder_q')
        self.state_dim, self.action_dim = state_dim, action_dim
        self.finalize_parameters()

    def get_embedding_input(self, state, lst_state, lst_action, reward) -> torch.Tensor:
        return _inputs.embedding_input(self, state, lst_state, lst_action, reward)

    def state_action(self, state, action, dest=None):
        return _inputs.encode_concat([(self.state_input_encoder, state), (self.action_input_encoder, action)],
                                     self.uni_model_input_mapping_activation_func if self.separate_encoder else None, dest=dest)

    def forward(self, state, lst_state, lst_action, action, rnn_memory: Optional[RNNHidden], reward, detach_embedding=False
                ) -> Tuple[torch.Tensor, torch.Tensor, RNNHidden, Optional[RNNHidden]]:
        emb_in = None if getattr(self, '_prefetched', None) is not None else self.get_embedding_input(state, lst_state, lst_action, reward)
        part = None
        # long GPU passes: the head input [state-action encoding | embedding] is ONE row buffer whose column blocks the producing GEMMs write
        # in place (no cat of 205 MB per pass at config 2, one shared magnitude handle instead of a pre-pass)
        rb = self.head_row_buffer(state.shape[:-1], state.device, state.dtype) if self.separate_encoder else None
        if detach_embedding and torch.is_grad_enabled() and action.requires_grad and self._action_only_graph():
            # actor step: frozen critic, detached embedding - the ONLY differentiable input of the head is the action encoding.
            # Encode state and action separately so that the first layer's backward forms just that column block of dX.
            act_fn = self.uni_model_input_mapping_activation_func
            n_s = self.state_input_encoder.weight.shape[0]
            with torch.no_grad():
                sa_s = _inputs.encode_concat([(self.state_input_encoder, state)], act_fn, dest=None if rb is None else rb.block(0, n_s))
            sa_a = _inputs.encode_concat([(self.action_input_encoder, action)], act_fn,
                                         dest=None if rb is None else rb.block(n_s, self.action_input_encoder.weight.shape[0]))
            sa = [(sa_s, 0), (sa_a.detach(), n_s)] if rb is not None else torch.cat((sa_s, sa_a.detach()), dim=-1)
            part = (sa_a, sa_s.shape[-1])
        else:
            sa = self.state_action(state, action, dest=None if rb is None else rb.block(0, rb.width - self.embedding_network.output_size))
        value, rnn_memory, emb, full = self.meta_forward(emb_in, sa, rnn_memory, detach_embedding, uni_grad_part=part, row_buffer=rb)
        return value, emb, rnn_memory, full

    def _action_only_graph(self) -> bool:
        import os
        if os.environ.get('RESEL_ACTION_ONLY_DX', '1') == '0':
            return False
        first = self.uni_network.layer_list[0]
        from ..models.ensemble_linear_model import EnsembleLinear
        return (self.separate_encoder and isinstance(self.state_input_encoder, torch.nn.Linear)
                and not self.state_input_encoder.weight.requires_grad and not self.action_input_encoder.weight.requires_grad
                and isinstance(first, EnsembleLinear) and not first.weight.requires_grad
                and isinstance(self.uni_network.activation_list[0], torch.nn.ELU) and len(self.uni_network.layer_list) > 1)

    def forward_embedding(self, state, lst_state, lst_action, rnn_memory, reward):
        return self.get_embedding(self.get_embedding_input(state, lst_state, lst_action, reward), rnn_memory)
