"""AdamW over a FlatParameterStore: one kernel per step, per-module learning rates (RESeL parameter groups,
reference algorithm/sac_full_length_rnn_redq_sep_optim.py:49-66), gradient scale read from device memory."""
import torch

from ..hip import ops
from ..models.flat_params import FlatParameterStore


class FlatAdamW:
    def __init__(self, store: FlatParameterStore, lr_of_module, wd_of_module, betas=(0.9, 0.999), eps=1e-8):
        self.store = store
        self.betas, self.eps = betas, eps
        self.lr_of_module, self.wd_of_module = lr_of_module, wd_of_module
        self.step_count = 0
        self._build()

    def _build(self):
        dev = self.store.flat.device
        self.m = torch.zeros(self.store.numel, dtype=torch.float32, device=dev)
        self.v = torch.zeros(self.store.numel, dtype=torch.float32, device=dev)
        self.seg_end, self.seg_lr, self.seg_wd = self.store.segments(self.lr_of_module, self.wd_of_module)

    def to(self, device):
        if self.m.device != torch.device(device):
            self.m, self.v = self.m.to(device), self.v.to(device)
            self.seg_end, self.seg_lr, self.seg_wd = self.seg_end.to(device), self.seg_lr.to(device), self.seg_wd.to(device)

    def zero_grad(self):
        self.store.zero_grad()

    def step(self, grad_scale: torch.Tensor = None):
        """grad_scale: optional 1-element device tensor multiplied into the gradient (e.g. 1 / global valid count)."""
        self.step_count += 1
        n = self.store.numel
        ops.adamw_flat_(self.store.flat[:n], self.store.grad[:n], self.m, self.v, self.seg_end, self.seg_lr, self.seg_wd,
                        self.step_count, self.betas[0], self.betas[1], self.eps, grad_scale)
