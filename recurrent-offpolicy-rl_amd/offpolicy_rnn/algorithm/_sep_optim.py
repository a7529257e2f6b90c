"""RESeL parameter groups: every sub-layer of the context encoder (`embedding_model`: input fc, sequence layer, output fc,
norms) trains at `rnn_*_lr`; input encoders, the universal MLP and the input mapping keep the base learning rate
(reference algorithm/sac_full_length_rnn_redq_sep_optim.py:49-66 - the "paper implementation" branch)."""
from .flat_adamw import FlatAdamW


def regroup(trainer):
    par = trainer.parameter
    trainer.optimizer_policy = FlatAdamW(trainer.policy.store,
                                         lambda m: par.rnn_policy_lr if m == 'embedding_model' else par.policy_lr,
                                         lambda m: par.policy_l2_norm)
    trainer.optimizers_value = [FlatAdamW(v.store, lambda m: par.rnn_value_lr if m == 'embedding_model' else par.value_lr,
                                          lambda m: par.value_l2_norm) for v in trainer.values]
