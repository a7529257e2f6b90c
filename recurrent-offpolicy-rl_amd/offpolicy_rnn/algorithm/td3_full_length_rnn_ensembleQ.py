"""TD3 twins of the full-trajectory trainers (reference offpolicy_rnn/algorithm/td3_full_length_rnn_ensembleQ.py:18-139):
deterministic actor, clipped-noise target smoothing, no entropy term, no alpha tuning."""
import torch

from ..utility import rng
from .sac_full_length_rnn_ensembleQ import SACFullLengthRNNEnsembleQ


class TD3FullLengthRNNEnsembleQ(SACFullLengthRNNEnsembleQ):
    target_from_live_policy = False       # this trainer smooths the (frozen) target policy; the REDQ one uses the live policy

    def __init__(self, parameter):
        super().__init__(parameter)
        self.parameter.no_alpha_auto_tune = True      # set after SAC.__init__ built log_alpha = 0, exactly as upstream

    def _target_action(self, mean, sample, logp):
        par = self.parameter
        noise = torch.clamp(rng.randn_like(mean) * par.target_action_noise_std, -par.target_action_noise_clip, par.target_action_noise_clip)
        return torch.clamp(mean + noise, -1, 1), None

    def _actor_objective(self, alpha, logp, q_pi):
        return -q_pi

    def _target_policy_update(self, tau):
        self.target_policy.copy_weight_from(self.policy, tau)
