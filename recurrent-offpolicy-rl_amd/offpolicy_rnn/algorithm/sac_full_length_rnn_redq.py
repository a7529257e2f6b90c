"""REDQ variant: target = min over a random subset of `redq_m` critics, actor loss uses the ensemble mean
(reference offpolicy_rnn/algorithm/sac_full_length_rnn_redq.py:16-49)."""
import numpy as np

from .sac_full_length_rnn_ensembleQ import SACFullLengthRNNEnsembleQ


class SACFullLengthRNNREDQ(SACFullLengthRNNEnsembleQ):
    def _select_target_ensemble(self, num_ensemble: int) -> np.ndarray:
        # the global numpy stream, draw for draw as the reference.  Under data parallelism every rank draws its own subset
        # for its own rows (independent REDQ samples per shard of the global batch); identically seeded ranks coincide.
        return np.random.permutation(num_ensemble)[:self.parameter.redq_m]

    def _q_for_policy(self, qs):
        return qs.mean(dim=0)
