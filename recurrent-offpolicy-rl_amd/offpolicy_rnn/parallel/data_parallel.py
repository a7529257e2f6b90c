"""Batch-row data parallelism: one process per GPU, identical parameters everywhere, each rank trains on its own
rows, ONE all-reduce (sum) of the flat gradient buffer per optimizer step (RCCL over xGMI; `gloo` in the CPU tests).

The buffer's trailing slots carry the rank-local valid-transition count (and the entropy-coefficient gradient for the
actor step), so the global normalisation `sum_r g_r / sum_r n_r` - exactly the reference's `/ valid_num`
(sac_full_length_rnn_ensembleQ.py:80-81,391) over the global batch - needs no second collective and no host sync.
The reference itself has no distributed code; this is new design."""
import os

import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        # `active`: the collectives are issued.  RESEL_DP_FORCE_COLLECTIVES=1 issues them in a one-rank group too (identity
        # results): the way to exercise the RCCL plumbing - side-stream all-reduce, broadcast, the guard's MAX all-reduces - on a
        # single-GPU box (`torchrun --nproc-per-node 1 bench.py --gpus 1`).
        self.active = self.world > 1 or (dist.is_available() and dist.is_initialized() and os.environ.get('RESEL_DP_FORCE_COLLECTIVES') == '1')
        self._stream, self._pending = None, None

    def all_reduce_(self, flat_grad: torch.Tensor):
        if self.active:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)

    def all_reduce_async_(self, flat_grad: torch.Tensor):
        """SUM all-reduce on a side stream (RCCL): returns at once, `wait()` makes the compute stream depend on its completion.
        gloo (CPU tests, or CUDA tensors staged through the host) and single-process runs reduce in place synchronously."""
        self._pending = None
        if not self.active:
            return
        if flat_grad.is_cuda and dist.get_backend(self.group) == 'nccl':
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=flat_grad.device)
            self._stream.wait_stream(torch.cuda.current_stream(flat_grad.device))
            with torch.cuda.stream(self._stream):
                dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)
            flat_grad.record_stream(self._stream)
            self._pending = flat_grad.device
        else:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)

    def wait(self):
        if getattr(self, '_pending', None) is not None:
            torch.cuda.current_stream(self._pending).wait_stream(self._stream)
            self._pending = None

    def broadcast_(self, flat: torch.Tensor, src=0):
        if self.active:
            dist.broadcast(flat, src=src, group=self.group)

    def all_reduce_max_(self, t: torch.Tensor):
        if self.active:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)


def init_from_env(backend=None):
    """torchrun-style initialisation (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local
