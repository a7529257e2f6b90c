"""Deterministic TD3 actor (reference offpolicy_rnn/policy_value_models/contextual_td3_policy.py:6-36)."""
import torch

from ..utility import rng
from .contextual_sac_policy import ContextualSACPolicy


class ContextualTD3Policy(ContextualSACPolicy):
    def __init__(self, state_dim, action_dim, embedding_size, embedding_hidden, embedding_activations, embedding_layer_type,
                 uni_model_hidden, uni_model_activations, uni_model_layer_type, fix_rnn_length, uni_model_input_mapping_dim: int = 0,
                 reward_input=False, last_action_input=True, last_state_input=False, separate_encoder=False, sample_std=0.1):
        super().__init__(state_dim, action_dim, embedding_size, embedding_hidden, embedding_activations, embedding_layer_type,
                         uni_model_hidden, uni_model_activations, uni_model_layer_type, fix_rnn_length, uni_model_input_mapping_dim,
                         reward_input, last_action_input, last_state_input, separate_encoder, output_logstd=False,
                         name='ContextualTD3Policy')
        self.sample_std = sample_std

    def forward(self, state, lst_state, lst_action, rnn_memory, reward=None, detach_embedding=False):
        emb_in = self.get_embedding_input(state, lst_state, lst_action, reward)
        out, rnn_memory, emb, full = self.meta_forward(emb_in, state, rnn_memory, detach_embedding)
        action_mean, action_sample, log_prob = self.process_model_out(out)
        return action_mean, emb, action_sample, log_prob, rnn_memory, full

    def process_model_out(self, out):
        """Head of the deterministic actor (reference :30-35): tanh mean, exploration sample with one N(0, I) draw."""
        action_mean = torch.tanh(out)
        action_sample = torch.clamp(action_mean + rng.randn_like(out) * self.sample_std, -1, 1)
        return action_mean, action_sample, torch.zeros_like(action_sample)
