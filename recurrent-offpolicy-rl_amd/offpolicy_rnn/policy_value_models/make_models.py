"""Model factory (reference offpolicy_rnn/policy_value_models/make_models.py:10-28)."""
from .contextual_sac_discrete_policy import ContextualSACDiscretePolicy
from .contextual_sac_discrete_value import ContextualSACDiscreteValue
from .contextual_sac_policy import ContextualSACPolicy
from .contextual_sac_value import ContextualSACValue
from .contextual_td3_policy import ContextualTD3Policy
from .contextual_td3_value import ContextualTD3Value


def make_policy_model(policy_args, base_alg_name, discrete):
    if base_alg_name == 'sac':
        return ContextualSACDiscretePolicy(**policy_args) if discrete else ContextualSACPolicy(**policy_args)
    if base_alg_name == 'td3':
        if discrete:
            raise NotImplementedError('TD3 has no discrete-action form (the reference only builds discrete SAC heads)')
        return ContextualTD3Policy(**policy_args)
    raise NotImplementedError(base_alg_name)


def make_value_model(value_args, base_alg_name, discrete):
    if base_alg_name == 'sac':
        return ContextualSACDiscreteValue(**value_args) if discrete else ContextualSACValue(**value_args)
    if base_alg_name == 'td3':
        if discrete:
            raise NotImplementedError('TD3 has no discrete-action form (the reference only builds discrete SAC heads)')
        return ContextualTD3Value(**value_args)
    raise NotImplementedError(base_alg_name)
