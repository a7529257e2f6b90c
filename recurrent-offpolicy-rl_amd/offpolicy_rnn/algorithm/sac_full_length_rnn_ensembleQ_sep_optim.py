from ._sep_optim import regroup
from .sac_full_length_rnn_ensembleQ import SACFullLengthRNNEnsembleQ


class SACFullLengthRNNENSEMBLEQ_SEP_OPTIM(SACFullLengthRNNEnsembleQ):
    def __init__(self, parameter):
        super().__init__(parameter)
        regroup(self)
        self.init_lr_scheduler()

    def init_lr_scheduler(self):
        pass
