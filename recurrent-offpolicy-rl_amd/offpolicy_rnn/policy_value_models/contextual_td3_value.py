from .contextual_sac_value import ContextualSACValue


class ContextualTD3Value(ContextualSACValue):
    def __init__(self, *args, **kwargs):
        kwargs.setdefault('name', 'ContextualTD3Value')
        super().__init__(*args, **kwargs)
