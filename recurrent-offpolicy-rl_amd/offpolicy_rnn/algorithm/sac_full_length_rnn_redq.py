"""REDQ variant: target = min over a random subset of `redq_m` critics, actor loss uses the ensemble mean
(reference offpolicy_rnn/algorithm/sac_full_length_rnn_redq.py:16-49)."""
import numpy as np

from .sac_full_length_rnn_ensembleQ import SACFullLengthRNNEnsembleQ


class SACFullLengthRNNREDQ(SACFullLengthRNNEnsembleQ):
    def _select_target_ensemble(self, num_ensemble: int) -> np.ndarray:
        # one process: the global numpy stream, draw for draw as the reference; data parallel: a stream shared by all ranks
        return self._subset_stream().permutation(num_ensemble)[:self.parameter.redq_m]

    actor_q_reduce = 'mean'

    def _q_for_policy(self, qs):
        return qs.mean(dim=0)
