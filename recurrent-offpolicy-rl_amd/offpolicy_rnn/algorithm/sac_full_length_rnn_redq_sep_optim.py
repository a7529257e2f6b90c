from ._sep_optim import regroup
from .sac_full_length_rnn_redq import SACFullLengthRNNREDQ


class SACFullLengthRNNREDQ_SEP_OPTIM(SACFullLengthRNNREDQ):
    """alg_name `sac_rnn_full_horizon_redQ_sep_optim` - the published RESeL configuration."""

    def __init__(self, parameter):
        super().__init__(parameter)
        regroup(self)
        self.init_lr_scheduler()

    def init_lr_scheduler(self):
        pass
