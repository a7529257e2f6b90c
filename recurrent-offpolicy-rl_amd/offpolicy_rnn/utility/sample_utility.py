"""numpy <-> torch helpers and action (un)normalisation (reference offpolicy_rnn/utility/sample_utility.py:18-36)."""
import numpy as np
import torch


def norm_act(act, act_space):
    if hasattr(act_space, 'low') and hasattr(act_space, 'high'):
        return (act - act_space.low) / (act_space.high - act_space.low) * 2 - 1
    return act


def unorm_act(act, act_space):
    if hasattr(act_space, 'low') and hasattr(act_space, 'high'):
        return (act + 1) / 2 * (act_space.high - act_space.low) + act_space.low
    return act


def n2t(data: np.ndarray, device) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(data)).to(torch.get_default_dtype()).to(device)


def n2t_2dim(data: np.ndarray, device) -> torch.Tensor:
    return n2t(data, device).reshape((-1, data.shape[-1]))


def t2n(data: torch.Tensor) -> np.ndarray:
    return data.detach().cpu().numpy()
