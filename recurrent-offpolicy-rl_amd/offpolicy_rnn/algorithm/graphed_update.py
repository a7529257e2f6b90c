"""One whole `train_one_batch()` as ONE hipGraph replay (small per-GPU batches are host-bound: about 500 kernel launches per update,
10 ms of wall time for 7.4 ms of kernels at 8 rows).  New design: the reference loops eagerly (algorithm/sac.py:359-362).

What varies from update to update is moved OUT of the captured region into static buffers that the host refreshes before each replay:
  * the sampling plan - the host half of `sample_trajs_device` (trajectory choice, numpy RNG, packing plan) runs as in the eager update
    and writes the int32 plan into a pinned buffer; the graph holds the H2D copy node and the gather kernel;
  * the REDQ critic subset - drawn on the host from the same numpy stream as the eager update, copied into static index tensors;
  * the step-dependent AdamW factors - `FlatAdamW.prepare_step()` (device-resident bias corrections), torch's `capturable` AdamW for
    the entropy coefficient;
  * the log scalars - a D2H node into a static pinned buffer, read on demand like `DeferredLog`.
Actor noise comes from torch's CUDA generator, which torch.cuda.graph registers: every replay draws fresh numbers.

A graph is valid for ONE batch shape (rows, row length, number of plan segments, longest segment) - synthetic benches, fixed-length
episodes.  Every `step()` is exactly ONE update (same update-to-data ratio and random streams as the eager loop): the first `warmup`
calls run eagerly (allocator and lazily initialised kernels warm up), a shape is recorded the SECOND time it occurs (a shape that
never recurs is not worth two device synchronisations and an activation pool), at most `max_graphs` graphs live at a time in ONE
shared memory pool (they never replay concurrently), the least recently used one is dropped for a newcomer; any update without a
graph runs eagerly from the same static inputs.  Refused at construction (use the eager `train_one_batch`): layers that need host-built
sequence tables or host dropout counters (cgpt), side-stream overlap (gru), gradient clipping, data-parallel groups."""
from collections import OrderedDict

import numpy as np
import torch

from ..hip import ops
from .sac_full_length_rnn_ensembleQ import DeferredLog


class _StaticLog(DeferredLog):
    def __init__(self, keys, buf, event, host_items):
        dict.__init__(self)
        self._keys, self._buf, self._event = keys, buf, event
        self.set_host(host_items)


class GraphedUpdate:
    PLAN_CAPACITY = 16384                             # plan segments the static buffers hold (64 KB of pinned memory)

    def __init__(self, alg, warmup=3, max_graphs=4):
        why = self.refusal(alg)
        if why:
            raise RuntimeError('GraphedUpdate: ' + why)
        self.alg, self.device = alg, alg.device
        self.graphs = OrderedDict()                   # batch shape key -> CUDAGraph, least recently used first
        self.warmup, self.max_graphs = warmup, max_graphs
        self._eager_left = warmup                     # updates still to run eagerly before anything is recorded
        self._seen = {}                               # batch shape key -> occurrences so far
        self._pool = None                             # memory pool shared by every recorded graph
        self.eager_fallbacks = 0
        E = alg.target_values[0].uni_network.layer_list[-1].num_ensemble
        self.E = E
        self._draw = type(alg)._select_target_ensemble.__get__(alg)         # the trainer's own host draw
        # static subset buffers sized WITHOUT consuming a draw: evaluate the draw on a scratch numpy stream and put the stream back
        st_np = np.random.get_state()
        rng_own = getattr(alg, '_subset_rng', None)
        st_own = None if rng_own is None else rng_own.get_state()
        first = np.asarray(self._draw(E))
        np.random.set_state(st_np)
        if st_own is not None:
            rng_own.set_state(st_own)
        self.subset_np = np.ascontiguousarray(first, dtype=np.int32)
        self.subset_i32 = torch.from_numpy(self.subset_np.copy()).to(self.device)
        self.subset_i64 = self.subset_i32.long()
        self._sub_host = torch.empty(self.subset_np.size, dtype=torch.int32, pin_memory=True)
        for opt in (alg.optimizer_value, alg.optimizer_policy):
            opt.enable_device_factors()
        oa = alg.optimizer_alpha                                             # torch AdamW over the entropy coefficient: capturable form
        for g in oa.param_groups:
            g['capturable'] = True
        for st in oa.state.values():
            if torch.is_tensor(st.get('step')):
                st['step'] = st['step'].to(self.device)
        self._plan_host = torch.empty((self.PLAN_CAPACITY, 4), dtype=torch.int32, pin_memory=True)
        self._plan_dev = torch.empty((self.PLAN_CAPACITY, 4), dtype=torch.int32, device=self.device)
        self._evt = torch.cuda.Event()
        self._log_host = torch.empty(64, dtype=torch.float32, pin_memory=True)      # static target of the log's D2H node
        self._last_log = None

    @staticmethod
    def refusal(alg):
        par = alg.parameter
        if alg.device.type != 'cuda':
            return 'needs a GPU'
        if getattr(alg, '_needs_seq_table', False):
            return 'attention layers build their sequence tables and dropout counters on the host'
        if getattr(alg, 'overlap_value_embedding', False):
            # tried in round 4: with the refusal lifted the capture of the gru trainer (target pass and prefetched value embeddings on
            # side streams, forked / joined with events) ends in a segmentation fault inside capture_end on this ROCm build
            return 'side-stream overlap (gru) is not captured'
        if getattr(alg, 'grad_sync', None) is None or alg.grad_sync.active:
            return 'data-parallel groups are not captured (or not a full-trajectory trainer)'
        if any(getattr(par, k, None) is not None for k in ('value_max_gradnorm', 'value_embedding_max_gradnorm', 'policy_max_gradnorm',
                                                           'policy_embedding_max_gradnorm')):
            return 'gradient clipping reads norms on the host'
        if par.utd != 1 or par.policy_update_per != 1 or par.randomize_mask or par.random_trunc_traj:
            return 'utd / policy_update_per != 1 or randomised masks change the launch sequence from update to update'
        if not (getattr(alg, 'device_replay', False) and alg.replay_buffer.device_supported(randomize_mask=par.randomize_mask)):
            return 'needs the device-resident replay ring'
        return None

    # ---- called by the trainer while `alg._graph is self` -------------------------------------------------------------------
    def gather(self):
        pl = self._plan
        n = pl['seg'].shape[0]
        self._plan_dev[:n].copy_(self._plan_host[:n], non_blocking=True)    # memcpy node: re-reads the pinned plan at every replay
        dev = self.alg.replay_buffer.gather_planned(self.device, self._plan_dev[:n], pl['max_len'], pl['nrow'], pl['longest'])
        return dev, pl['total_size'], pl['table']

    def log_node(self, keys, packed, host, items):
        self._log_keys = keys
        self._log_host[:packed.numel()].copy_(packed, non_blocking=True)
        self._log_items = dict(host, **items)
        return None

    # ---- host side of one update ----------------------------------------------------------------------------------------------
    def _prepare(self):
        alg, par = self.alg, self.alg.parameter
        pl = alg.replay_buffer.plan_trajs_device(par.sac_batch_size, None, random_trunc_traj=par.random_trunc_traj,
                                                 nest_stack_trajs=alg.allow_nest_stack)
        alg.replay_buffer._mirror(self.device)        # transitions pushed since the last update reach the device ring HERE, outside the graph
        n = pl['seg'].shape[0]
        if n > self.PLAN_CAPACITY:
            raise RuntimeError(f'GraphedUpdate: {n} plan segments exceed the static plan buffer ({self.PLAN_CAPACITY})')
        key = (pl['nrow'], pl['longest'], pl['max_len'], n)
        self._evt.synchronize()                                              # the previous update has read the pinned plan
        self._plan_host[:n].copy_(torch.from_numpy(pl['seg']))
        self._plan = pl
        sub = np.ascontiguousarray(np.asarray(self._draw(self.E)), dtype=np.int32)
        self._sub_host.copy_(torch.from_numpy(sub))
        self.subset_i32.copy_(self._sub_host, non_blocking=True)
        self.subset_i64.copy_(self.subset_i32)
        for opt in (alg.optimizer_value, alg.optimizer_policy):
            opt.prepare_step()
        return key

    def _body(self):
        alg = self.alg
        alg._graph = self
        alg._select_target_ensemble = lambda E: self.subset_np
        for opt in (alg.optimizer_value, alg.optimizer_policy):
            opt.device_factors_active = True
        try:
            ops.amax_arena_zero(self.device)          # first node: a replay publishes operand magnitudes into the slots / epochs baked into the graph
            alg.train_one_batch()
        finally:
            alg._graph = None
            del alg._select_target_ensemble
            for opt in (alg.optimizer_value, alg.optimizer_policy):
                opt.device_factors_active = False

    def _finish(self):
        self._evt.record()
        n = len(self._log_keys)
        items = dict(self._log_items)
        pl = self._plan                               # the host entries of THIS update (the captured dict holds the recorded update's)
        items.update(real_batch_size=pl['total_size'], real_batch_traj_num=pl['nrow'],
                     average_traj_len=self.alg.replay_buffer.size / len(self.alg.replay_buffer))
        self._last_log = _StaticLog(list(self._log_keys), self._log_host[:n], self._evt, items)
        return self._last_log

    def step(self):
        """Exactly one update: eager while warming up or while the batch shape has no graph, a replay otherwise."""
        if self._last_log is not None:
            self._last_log.resolve()                  # its pinned buffer is about to be rewritten
        key = self._prepare()
        g = self.graphs.get(key)
        if g is None:
            seen = self._seen[key] = self._seen.get(key, 0) + 1
            if self._eager_left > 0 or seen < 2:      # warm-up updates / a shape on its first visit: the same update, launched eagerly
                self._eager_left = max(0, self._eager_left - 1)
                self.eager_fallbacks += 1
                self._body()
                return self._finish()
            if len(self.graphs) >= self.max_graphs:
                self.graphs.popitem(last=False)       # least recently used; its activations return to the shared pool
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=self._pool):
                self._body()                          # recorded, not run: the prepared inputs are consumed by the replay below
            if self._pool is None:
                self._pool = g.pool()
            torch.cuda.synchronize(self.device)
            self.graphs[key] = g
        self.graphs.move_to_end(key)
        g.replay()
        return self._finish()

    @property
    def graph(self):                                  # the most recently recorded graph (tests)
        return next(reversed(self.graphs.values()), None) if self.graphs else None
