"""Input encoders shared by actor and critic: `[state, last_state, last_action, reward]` filtered by flags, each
through its own Linear(., 128) when `separate_encoder` (reference contextual_sac_value.py:27-48,90-99)."""
import torch

BASIC_EMBEDDING_DIM = 128


def build_encoders(owner, state_dim, action_dim, reward_input, last_action_input, last_state_input, separate_encoder):
    owner.reward_input, owner.last_action_input, owner.last_state_input = reward_input, last_action_input, last_state_input
    owner.reward_dim = 1 if reward_input else 0
    owner.last_act_dim = action_dim if last_action_input else 0
    owner.last_obs_dim = state_dim if last_state_input else 0
    owner.separate_encoder = separate_encoder
    if separate_encoder:
        d = BASIC_EMBEDDING_DIM
        owner.state_encoder = torch.nn.Linear(state_dim, d)
        owner.last_act_encoder = torch.nn.Linear(owner.last_act_dim, d) if owner.last_act_dim else None
        owner.reward_encoder = torch.nn.Linear(owner.reward_dim, d) if owner.reward_dim else None
        owner.last_obs_encoder = torch.nn.Linear(owner.last_obs_dim, d) if owner.last_obs_dim else None
        return d * (1 + sum(e is not None for e in (owner.last_act_encoder, owner.last_obs_encoder, owner.reward_encoder)))
    ident = torch.nn.Identity()
    owner.state_encoder = owner.last_act_encoder = owner.reward_encoder = owner.last_obs_encoder = ident
    return state_dim + owner.reward_dim + owner.last_act_dim + owner.last_obs_dim


def register_encoders(owner):
    if not owner.separate_encoder:
        return
    owner.contextual_register_rnn_base_module(owner.state_encoder, 'state_encoder')
    for mod, name in ((owner.last_act_encoder, 'last_act_encoder'), (owner.last_obs_encoder, 'last_obs_encoder'),
                      (owner.reward_encoder, 'reward_encoder')):
        if mod is not None:
            owner.contextual_register_rnn_base_module(mod, name)


def embedding_input(owner, state, lst_state, lst_action, reward) -> torch.Tensor:
    parts = [owner.state_encoder(state)]
    if owner.last_state_input:
        parts.append(owner.last_obs_encoder(lst_state))
    if owner.last_action_input:
        parts.append(owner.last_act_encoder(lst_action))
    if owner.reward_input:
        parts.append(owner.reward_encoder(reward))
    return torch.cat(parts, dim=-1)
