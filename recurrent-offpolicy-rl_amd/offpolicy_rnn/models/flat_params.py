"""One contiguous fp32 buffer per network (MI355X-first parameter layout).

Every nn.Parameter of a ContextualModel becomes a view into `flat`, every `.grad` a view into `grad`, in the
contractual module / parameter order.  That turns the per-tensor Python loops of the
reference into single HBM-bound kernels:
    soft target update (rnn_base.py:490-491)        -> ops.soft_update_(target.flat, online.flat, tau)
    AdamW with RESeL lr groups (…_sep_optim.py:49-66) -> ops.adamw_flat_(flat, grad, m, v, segment table)
    l2_norm_square (rnn_base.py:531-532)            -> ops.sumsq(flat[prefix])
    data-parallel gradient exchange                 -> ONE RCCL all-reduce of `grad`, no bucketing copies.
Each tensor starts on a 16-byte boundary (kernels take float4 loads of D / delta_bias / norm weights)."""
from collections import OrderedDict
from typing import Dict, List, Tuple

import torch


class FlatParameterStore:
    ALIGN = 4          # elements

    def __init__(self, modules: 'OrderedDict[str, torch.nn.Module]', extra: int = 4):
        """`extra` trailing fp32 slots ride along with the gradient buffer (e.g. the local valid-transition count and
        the entropy-coefficient gradient), so that one all-reduce carries everything a step needs."""
        self.modules = modules
        self.slices: List[Tuple[torch.nn.Parameter, int, int]] = []
        self.module_range: Dict[str, Tuple[int, int]] = {}
        off = 0
        for name, mod in modules.items():
            begin = off
            for p in mod.parameters(True):
                n = p.numel()
                self.slices.append((p, off, n))
                off += (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
            self.module_range[name] = (begin, off)
        self.numel = off
        self.extra = extra
        dev = self.slices[0][0].device if self.slices else torch.device('cpu')
        self.flat = torch.zeros(off + extra, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off + extra, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o, n in self.slices:
                self.flat[o:o + n].copy_(p.detach().reshape(-1))
        self._amax = None            # per-tensor magnitude handles (GEMM product mode 2): (handles, PARAM_EPOCH at refresh, versions)
        self._repoint()
        from ..hip import ops
        ops.register_store(self)     # `ops.amax_maintenance` drops the weight handles of every store when the epoch counter starts over

    def _repoint(self):
        for i, (p, o, n) in enumerate(self.slices):
            p.data = self.flat[o:o + n].view(p.shape)
            p.grad = self.grad[o:o + n].view(p.shape)
            p._resel_store = (self, i)
        self._amax = None

    def amax_handle(self, index: int, p):
        """Magnitude handle of parameter `index` (hip/ops.py `weight_amax`).  All tensors of the buffer are refreshed by ONE launch when
        the buffer was rewritten since the last refresh: by this library's in-place kernels (ops.PARAM_EPOCH) or by an in-place torch op
        on any of the parameters (their `_version`s; load_state_dict, initialisers)."""
        from ..hip import ops
        if self.flat.device.type != 'cuda' or p.data_ptr() != self.flat.data_ptr() + 4 * self.slices[index][1]:
            return None                                   # not (or no longer) a view of this buffer
        st = self._amax
        if st is None or st[1] != ops.PARAM_EPOCH[0] or st[2][index] != p._version or st[0].device != self.flat.device:
            vers = [q._version for q, _, _ in self.slices]
            if st is None or st[0].device != self.flat.device:
                dev = self.flat.device
                handles = torch.zeros(len(self.slices) * ops.AMAX_WORDS, dtype=torch.int64, device=dev)
                begin = torch.tensor([o for _, o, _ in self.slices], dtype=torch.int64, device=dev)
                length = torch.tensor([n for _, _, n in self.slices], dtype=torch.int64, device=dev)
                views = list(handles.view(len(self.slices), ops.AMAX_WORDS).unbind(0))
            else:
                handles, begin, length, views = st[0], st[3], st[4], st[5]
            if torch.cuda.is_current_stream_capturing():
                handles.zero_()                           # a replay publishes with the epoch baked into the node: start from nothing, not from the last replay's maxima
            ops.amax_segments(self.flat, begin, length, handles)
            st = self._amax = (handles, ops.PARAM_EPOCH[0], vers, begin, length, views)
        return st[5][index]

    def ensure_grad_extra(self, n: int):
        """At least n trailing slots behind the gradients (the parameter buffer keeps its own tail): data-parallel runs carry every
        rank's Q-guard extrema there, four floats per rank, behind the four standing slots."""
        have = self.grad.numel() - self.numel
        if have >= n:
            return
        grad = torch.zeros(self.numel + n, dtype=torch.float32, device=self.grad.device)
        grad[:self.grad.numel()].copy_(self.grad)
        self.grad = grad
        for p, o, k in self.slices:
            if p.grad is not None:
                p.grad = self.grad[o:o + k].view(p.shape)

    def to(self, device):
        if self.flat.device != torch.device(device):
            self.flat = self.flat.to(device)
            self.grad = self.grad.to(device)
            self._repoint()
        return self

    def zero_grad(self):
        """Zero the flat gradient buffer and detach the per-parameter `.grad`s: with `.grad = None` autograd's AccumulateGrad
        keeps the incoming gradient tensor as is (no kernel) instead of launching one tiny add per parameter into the view;
        `collect_grads()` then moves all of them into the flat buffer with a multi-tensor copy (2-3 launches per network
        instead of ~40)."""
        self.grad.zero_()
        for p, _, _ in self.slices:
            p.grad = None

    def collect_grads(self):
        """After backward(): gather the gradients autograd left on the parameters into the flat buffer and re-point
        `.grad` at the views (so that everything downstream - clipping, all-reduce, AdamW, user code - sees one buffer)."""
        dst, src = [], []
        base = self.grad.data_ptr()
        for p, o, n in self.slices:
            g = p.grad
            view = self.grad[o:o + n].view(p.shape)
            if g is not None and g.data_ptr() != base + 4 * o:
                dst.append(view)
                src.append(g.detach())
            p.grad = view
        if dst:
            torch._foreach_copy_(dst, src)

    def segments(self, lr_of_module, wd_of_module):
        """-> (seg_end int64[n], seg_lr[n], seg_wd[n]) tensors on the buffer's device, one segment per module."""
        ends, lrs, wds = [], [], []
        for name, (a, b) in self.module_range.items():
            if b > a:
                ends.append(b)
                lrs.append(lr_of_module(name))
                wds.append(wd_of_module(name))
        dev = self.flat.device
        return (torch.tensor(ends, dtype=torch.int64, device=dev), torch.tensor(lrs, dtype=torch.float32, device=dev),
                torch.tensor(wds, dtype=torch.float32, device=dev))
