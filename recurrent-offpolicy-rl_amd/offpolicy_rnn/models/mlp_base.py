"""MLPBase: an all-`fc` RNNBase used for the universal-model input mapping (reference offpolicy_rnn/models/mlp_base.py)."""
from .RNNHidden import RNNHidden
from .rnn_base import RNNBase


class MLPBase(RNNBase):
    def __init__(self, input_size, output_size, hidden_size_list, activation):
        super().__init__(input_size, output_size, hidden_size_list, activation, ['fc'] * len(activation))
        self.empty_hidden_state = RNNHidden(0, [])

    def meta_forward(self, x, h=None, require_full_hidden=False, out_dest=None):
        return super().meta_forward(x, self.empty_hidden_state, False, out_dest=out_dest)[0]

    def forward(self, x, out_dest=None):
        return self.meta_forward(x, out_dest=out_dest)
