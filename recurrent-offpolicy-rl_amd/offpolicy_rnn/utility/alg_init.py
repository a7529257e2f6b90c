"""alg_name -> trainer class (reference offpolicy_rnn/utility/alg_init.py:16-47).

Only the full-trajectory recurrent family is in scope of this build; the remaining reference
names raise NotImplementedError with an explicit message instead of silently mapping to something else."""

_OUT_OF_SCOPE = ('sac_no_train', 'sac_mlp', 'sac_mlp_redq', 'sac_rnn_slice', 'sac_mlp_redq_ensemble_q')


def alg_init(parameter):
    from ..algorithm.sac_full_length_rnn_ensembleQ import SACFullLengthRNNEnsembleQ
    from ..algorithm.sac_full_length_rnn_redq import SACFullLengthRNNREDQ
    from ..algorithm.sac_full_length_rnn_redq_sep_optim import SACFullLengthRNNREDQ_SEP_OPTIM
    from ..algorithm.sac_full_length_rnn_ensembleQ_sep_optim import SACFullLengthRNNENSEMBLEQ_SEP_OPTIM
    from ..algorithm.td3_full_length_rnn_ensembleQ import TD3FullLengthRNNEnsembleQ
    from ..algorithm.td3_full_length_rnn_redq import TD3FullLengthRNNREDQ
    from ..algorithm.td3_full_length_rnn_redq_sep_optim import TD3FullLengthRNNREDQ_SEP_OPTIM
    table = {
        'sac_rnn_full_horizon_ensembleQ': (SACFullLengthRNNEnsembleQ, 'sac'),
        'sac_rnn_full_horizon_redQ': (SACFullLengthRNNREDQ, 'sac'),
        'sac_rnn_full_horizon_redQ_sep_optim': (SACFullLengthRNNREDQ_SEP_OPTIM, 'sac'),
        'sac_rnn_full_horizon_ensemble_q_sep_optim': (SACFullLengthRNNENSEMBLEQ_SEP_OPTIM, 'sac'),
        'td3_rnn_full_horizon_ensembleQ': (TD3FullLengthRNNEnsembleQ, 'td3'),
        'td3_rnn_full_horizon_redQ': (TD3FullLengthRNNREDQ, 'td3'),
        'td3_rnn_full_horizon_redQ_sep_optim': (TD3FullLengthRNNREDQ_SEP_OPTIM, 'td3'),
    }
    name = parameter.alg_name
    if name in table:
        cls, base = table[name]
        if base == 'td3':
            parameter.base_algorithm = 'td3'
        return cls(parameter)
    if name in _OUT_OF_SCOPE:
        raise NotImplementedError(f'Algorithm {name} is a transition-level / slice trainer of the reference and is outside '
                                  f'the MI355X full-trajectory hot path implemented here.')
    raise NotImplementedError(f'Algorithm {name} has not been implemented!')
