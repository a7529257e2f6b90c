from ._sep_optim import regroup
from .td3_full_length_rnn_redq import TD3FullLengthRNNREDQ


class TD3FullLengthRNNREDQ_SEP_OPTIM(TD3FullLengthRNNREDQ):
    def __init__(self, parameter):
        super().__init__(parameter)
        regroup(self)
        self.init_lr_scheduler()

    def init_lr_scheduler(self):
        pass
