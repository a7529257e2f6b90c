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

    # ---- captured updates (algorithm/graphed_update.py): the step-dependent factors live in device memory and are refreshed by the
    # host before every replay; `step()` then launches the kernel that reads them and counts nothing itself
    def enable_device_factors(self):
        dev = self.store.flat.device
        self._bc_host = torch.empty(2, dtype=torch.float32, pin_memory=True)
        self._bc_dev = torch.zeros(2, dtype=torch.float32, device=dev)

    def prepare_step(self, stage=None):
        """Host side of one captured step: count it and send (1 - beta1^t, sqrt(1 - beta2^t)) to the device (stream-ordered copy).
        `stage`: a pinned 2-float block that no queued copy reads any more (a caller that runs ahead of the device hands out its own)."""
        self.step_count += 1
        t = float(self.step_count)
        host = self._bc_host if stage is None else stage
        host[0] = 1.0 - self.betas[0] ** t
        host[1] = (1.0 - self.betas[1] ** t) ** 0.5
        self._bc_dev.copy_(host, non_blocking=True)

    def step(self, grad_scale: torch.Tensor = None):
        """grad_scale: optional 1-element device tensor multiplied into the gradient (e.g. 1 / global valid count)."""
        n = self.store.numel
        if getattr(self, '_bc_dev', None) is not None and getattr(self, 'device_factors_active', False):
            ops.adamw_flat_dev_(self.store.flat[:n], self.store.grad[:n], self.m, self.v, self.seg_end, self.seg_lr, self.seg_wd,
                                self._bc_dev, self.betas[0], self.betas[1], self.eps, grad_scale)
            return
        self.step_count += 1
        ops.adamw_flat_(self.store.flat[:n], self.store.grad[:n], self.m, self.v, self.seg_end, self.seg_lr, self.seg_wd,
                        self.step_count, self.betas[0], self.betas[1], self.eps, grad_scale)
