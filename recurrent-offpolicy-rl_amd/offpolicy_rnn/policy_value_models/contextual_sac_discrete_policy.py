"""Categorical actor for discrete action spaces (reference
offpolicy_rnn/policy_value_models/contextual_sac_discrete_policy.py:13-138): logits -> softmax, mixed with a 0.01 floor
and renormalised; returns (argmax, sample, log-probabilities of ALL actions)."""
from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from ..models.RNNHidden import RNNHidden
from ..models.contextual_model import ContextualModel
from . import _inputs
from .utils import nearest_power_of_two, nearest_power_of_two_half


class ContextualSACDiscretePolicy(ContextualModel):
    MAX_LOG_STD = 2.0
    MIN_LOG_STD = -15.0

    def __init__(self, state_dim, action_dim, embedding_size, embedding_hidden, embedding_activations, embedding_layer_type,
                 uni_model_hidden, uni_model_activations, uni_model_layer_type, fix_rnn_length, uni_model_input_mapping_dim: int = 0,
                 reward_input=False, last_action_input=True, last_state_input=False, separate_encoder=False):
        if uni_model_activations[-1] != 'linear':
            uni_model_activations = list(uni_model_activations[:-1]) + ['linear']
        if embedding_size == 'auto':
            embedding_size = nearest_power_of_two_half(state_dim)
        if uni_model_input_mapping_dim == 'auto':
            uni_model_input_mapping_dim = nearest_power_of_two(state_dim)
        cum_dim = _inputs.build_encoders(self, state_dim, action_dim, reward_input, last_action_input, last_state_input, separate_encoder)
        super().__init__(embedding_input_size=cum_dim, embedding_size=embedding_size, embedding_hidden=embedding_hidden,
                         embedding_activations=embedding_activations, embedding_layer_type=embedding_layer_type,
                         uni_model_input_size=state_dim, uni_model_output_size=action_dim, uni_model_hidden=uni_model_hidden,
                         uni_model_activations=uni_model_activations, uni_model_layer_type=uni_model_layer_type,
                         fix_rnn_length=fix_rnn_length, uni_model_input_mapping_dim=uni_model_input_mapping_dim,
                         uni_model_input_mapping_activation=embedding_activations[-1], name='ContextualSACDiscretePolicy')
        _inputs.register_encoders(self)
        self.state_dim, self.action_dim = state_dim, action_dim
        self.finalize_parameters()

    def get_embedding_input(self, state, lst_state, lst_action, reward) -> torch.Tensor:
        return _inputs.embedding_input(self, state, lst_state, lst_action, reward)

    def forward(self, state, lst_state, lst_action, rnn_memory: Optional[RNNHidden], reward=None, detach_embedding=False
                ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, RNNHidden, Optional[RNNHidden]]:
        emb_in = self.get_embedding_input(state, lst_state, lst_action, reward)
        out, rnn_memory, emb, full = self.meta_forward(emb_in, state, rnn_memory, detach_embedding)
        action_mean, action_sample, log_probs, _ = self.process_model_out(out)
        return action_mean, emb, action_sample, log_probs, rnn_memory, full

    def process_model_out(self, model_output):
        probs = torch.softmax(model_output, dim=-1)                  # reference :112-113 (max-shifted exp / sum)
        probs = probs + 0.01
        probs = probs / probs.sum(dim=-1, keepdim=True)
        probs = probs / probs.sum(dim=-1, keepdim=True)              # torch.distributions.Categorical normalises once more (:116)
        action_mean = probs.argmax(dim=-1, keepdim=True)             # Categorical.mode
        action_sample = torch.multinomial(probs.reshape(-1, probs.shape[-1]), 1, True).reshape(probs.shape[:-1] + (1,))
        return action_mean, action_sample, torch.log(probs), probs

    def select_with_action(self, action: torch.Tensor, data: torch.Tensor) -> torch.Tensor:
        return data.gather(-1, action.long())

    def action2onehot(self, action: torch.Tensor):
        return F.one_hot(action.squeeze(-1).long(), num_classes=self.action_dim).float()

    def forward_embedding(self, state, lst_state, lst_action, rnn_memory, reward):
        return self.get_embedding(self.get_embedding_input(state, lst_state, lst_action, reward), rnn_memory)
