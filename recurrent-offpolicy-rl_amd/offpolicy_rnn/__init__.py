"""MI355X-native `offpolicy_rnn`: drop-in API surface of FanmingL/Recurrent-Offpolicy-RL for the full-trajectory
recurrent SAC/TD3 update (`from offpolicy_rnn import init_smart_logger, Parameter, alg_init`, reference
offpolicy_rnn/__init__.py:3,4,11).  Attributes are resolved lazily so that importing a sub-module (e.g. the HIP ops)
does not pull the trainer stack.
"""
_LAZY = {
    'init_smart_logger': ('.config.load_config', 'init_smart_logger'),
    'Parameter': ('.parameter.ParameterSAC', 'Parameter'),
    'alg_init': ('.utility.alg_init', 'alg_init'),
    'SAC': ('.algorithm.sac', 'SAC'),
    'SACFullLengthRNNEnsembleQ': ('.algorithm.sac_full_length_rnn_ensembleQ', 'SACFullLengthRNNEnsembleQ'),
    'SACFullLengthRNNREDQ': ('.algorithm.sac_full_length_rnn_redq', 'SACFullLengthRNNREDQ'),
    'SACFullLengthRNNREDQ_SEP_OPTIM': ('.algorithm.sac_full_length_rnn_redq_sep_optim', 'SACFullLengthRNNREDQ_SEP_OPTIM'),
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        mod, attr = _LAZY[name]
        return getattr(importlib.import_module(mod, __name__), attr)
    raise AttributeError(f'module {__name__!r} has no attribute {name!r}')
