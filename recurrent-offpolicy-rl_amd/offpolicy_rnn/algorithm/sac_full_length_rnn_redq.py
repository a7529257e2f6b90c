"""REDQ variant: target = min over a random subset of `redq_m` critics, actor loss uses the ensemble mean
(reference offpolicy_rnn/algorithm/sac_full_length_rnn_redq.py:16-49)."""
import numpy as np

from .sac_full_length_rnn_ensembleQ import SACFullLengthRNNEnsembleQ


class SACFullLengthRNNREDQ(SACFullLengthRNNEnsembleQ):
    def _select_target_ensemble(self, num_ensemble: int) -> np.ndarray:
        return np.random.permutation(num_ensemble)[:self.parameter.redq_m]      # host RNG: identical on every DP rank

    def _q_for_policy(self, qs):
        return qs.mean(dim=0)
