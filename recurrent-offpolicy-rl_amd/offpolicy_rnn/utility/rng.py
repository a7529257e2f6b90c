"""Single entry point for the Gaussian noise of the actor head / TD3 smoothing (reference: `torch.randn_like`,
contextual_sac_policy_single_head.py:114, contextual_td3_policy.py:32, td3_full_length_rnn_redq.py:22), so that parity
tests can feed the same draws to the CPU oracle and to the GPU trainer."""
import torch


def randn(shape, device, dtype=torch.float32):
    return torch.randn(tuple(shape), device=device, dtype=dtype)


def randn_like(t: torch.Tensor):
    return randn(t.shape, t.device, t.dtype)
