"""Model factory (reference offpolicy_rnn/policy_value_models/make_models.py:10-28).  Discrete-action heads are outside
the continuous-control hot path of this build."""
from .contextual_sac_policy import ContextualSACPolicy
from .contextual_sac_value import ContextualSACValue
from .contextual_td3_policy import ContextualTD3Policy
from .contextual_td3_value import ContextualTD3Value


def _no_discrete(discrete):
    if discrete:
        raise NotImplementedError('discrete-action actor / critic heads are outside the MI355X hot path of this build')


def make_policy_model(policy_args, base_alg_name, discrete):
    _no_discrete(discrete)
    if base_alg_name == 'sac':
        return ContextualSACPolicy(**policy_args)
    if base_alg_name == 'td3':
        return ContextualTD3Policy(**policy_args)
    raise NotImplementedError(base_alg_name)


def make_value_model(value_args, base_alg_name, discrete):
    _no_discrete(discrete)
    if base_alg_name == 'sac':
        return ContextualSACValue(**value_args)
    if base_alg_name == 'td3':
        return ContextualTD3Value(**value_args)
    raise NotImplementedError(base_alg_name)
