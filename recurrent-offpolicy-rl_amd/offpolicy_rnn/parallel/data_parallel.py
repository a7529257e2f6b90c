"""Batch-row data parallelism: one process per GPU, identical parameters everywhere, each rank trains on its own
rows, ONE all-reduce (sum) of the flat gradient buffer per optimizer step (RCCL over xGMI; `gloo` in the CPU tests).

The buffer's trailing slots carry the rank-local valid-transition count (and the entropy-coefficient gradient for the
actor step), so the global normalisation `sum_r g_r / sum_r n_r` - exactly the reference's `/ valid_num`
(sac_full_length_rnn_ensembleQ.py:80-81,391) over the global batch - needs no second collective and no host sync.
The reference itself has no distributed code (it launches independent seeds per GPU,
gen_tmuxp_mamba_pomdp.py:30-38); this is new design.

Every collective issued is counted per kind (`GradSync.calls`) together with the bytes it moved, so that a bench line or a
test can PROVE that the exchange ran (and on which backend) instead of inferring it from a finished run."""
import os
import socket

import torch
import torch.distributed as dist


def force_collectives():
    """RESEL_DP_FORCE_COLLECTIVES=1: issue the collectives in a one-rank group too (identity results) - the way to run the
    RCCL plumbing (communicator creation, side-stream all-reduce, broadcast, the guard's MAX all-reduces) on a one-GPU box."""
    return os.environ.get('RESEL_DP_FORCE_COLLECTIVES') == '1'


class GradSync:
    def __init__(self, group=None):
        self.group = group
        self.initialized = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.initialized else 1
        self.rank = dist.get_rank(group) if self.initialized else 0
        self.backend = dist.get_backend(group) if self.initialized else None
        self.active = self.world > 1 or (self.initialized and force_collectives())
        self._stream, self._pending = None, None
        self.calls = dict(all_reduce_sum=0, all_reduce_max=0, broadcast=0)
        self.bytes = dict(all_reduce_sum=0, all_reduce_max=0, broadcast=0)

    def _count(self, kind, t):
        self.calls[kind] += 1
        self.bytes[kind] += t.numel() * t.element_size()

    def reset_counters(self):
        for k in self.calls:
            self.calls[k] = self.bytes[k] = 0

    def all_reduce_(self, flat_grad: torch.Tensor):
        if self.active:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)
            self._count('all_reduce_sum', flat_grad)

    def all_reduce_async_(self, flat_grad: torch.Tensor):
        """SUM all-reduce on a side stream (RCCL): returns at once, `wait()` makes the compute stream depend on its completion.
        gloo (CPU tests, or CUDA tensors staged through the host) and single-process runs reduce in place synchronously."""
        self._pending = None
        if not self.active:
            return
        if flat_grad.is_cuda and self.backend == 'nccl':
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=flat_grad.device)
            self._stream.wait_stream(torch.cuda.current_stream(flat_grad.device))
            with torch.cuda.stream(self._stream):
                dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)
            flat_grad.record_stream(self._stream)
            self._pending = flat_grad.device
        else:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)
        self._count('all_reduce_sum', flat_grad)

    def wait(self):
        if getattr(self, '_pending', None) is not None:
            torch.cuda.current_stream(self._pending).wait_stream(self._stream)
            self._pending = None

    def broadcast_(self, flat: torch.Tensor, src=0):
        if self.active:
            dist.broadcast(flat, src=src, group=self.group)
            self._count('broadcast', flat)
            if flat.is_cuda:                         # a raw write into a parameter buffer: the weight magnitudes of GEMM mode 2 are stale now
                from ..hip import ops
                ops.PARAM_EPOCH[0] += 1

    def all_reduce_max_(self, t: torch.Tensor):
        if self.active:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            self._count('all_reduce_max', t)


def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def init_from_env(backend=None):
    """torchrun-style initialisation (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).  Returns (rank, world, local_rank).
    A group is created when WORLD_SIZE > 1, and also for ONE rank when RESEL_DP_FORCE_COLLECTIVES=1 (MASTER_* default to a
    free local port then), so that a one-GPU box runs the same RCCL calls an N-GPU job does."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if (world > 1 or force_collectives()) and not dist.is_initialized():
        if backend is None:                          # RESEL_DP_BACKEND=gloo: several ranks may then share one GPU (tests on a one-GPU box)
            backend = os.environ.get('RESEL_DP_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if world == 1:
            os.environ.setdefault('MASTER_PORT', str(free_port()))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local
