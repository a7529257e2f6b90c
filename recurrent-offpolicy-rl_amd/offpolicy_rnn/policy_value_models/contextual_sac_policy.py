from .contextual_sac_policy_single_head import ContextualSACPolicySingleHead


class ContextualSACPolicy(ContextualSACPolicySingleHead):
    """Alias kept for API compatibility (reference contextual_sac_policy.py:4-15)."""
