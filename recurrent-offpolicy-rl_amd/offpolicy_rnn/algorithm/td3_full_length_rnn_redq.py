import numpy as np

from .td3_full_length_rnn_ensembleQ import TD3FullLengthRNNEnsembleQ


class TD3FullLengthRNNREDQ(TD3FullLengthRNNEnsembleQ):
    """reference offpolicy_rnn/algorithm/td3_full_length_rnn_redq.py:14-51"""
    target_from_live_policy = True

    def _select_target_ensemble(self, num_ensemble: int) -> np.ndarray:
        return self._subset_stream().permutation(num_ensemble)[:self.parameter.redq_m]

    actor_q_reduce = 'mean'

    def _q_for_policy(self, qs):
        return qs.mean(dim=0)
