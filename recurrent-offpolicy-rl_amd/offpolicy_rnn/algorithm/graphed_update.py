"""One whole `train_one_batch()` as ONE hipGraph replay (small per-GPU batches are host-bound: about 500 kernel launches per update,
10 ms of wall time for 7.4 ms of kernels at 8 rows).  New design: the reference loops eagerly (algorithm/sac.py:359-362).

What varies from update to update is moved OUT of the captured region into static buffers that the host refreshes before each replay:
  * the sampling plan - the host half of `sample_trajs_device` (trajectory choice, numpy RNG, packing plan) runs as in the eager update
    and writes the int32 plan into a pinned buffer; the graph holds the H2D copy node and the gather kernel;
  * the REDQ critic subset - drawn on the host from the same numpy stream as the eager update, copied into static index tensors;
  * the step-dependent AdamW factors - `FlatAdamW.prepare_step()` (device-resident bias corrections), torch's `capturable` AdamW for
    the entropy coefficient;
  * the log scalars - packed into a static device buffer by the graph, copied to pinned memory behind the replay, read on demand like
    `DeferredLog`.
Every host -> device refresh and the log's read-back go through a RING of pinned staging slots and run as ordinary stream-ordered
copies around the replay (not as nodes of it): `step()` never waits for the update it has just launched - the host prepares update
i + 1 (sampling plan, tables, graph launch) while update i runs; a slot is reused three updates later, after its event.
Actor noise comes from torch's CUDA generator, which torch.cuda.graph registers: every replay draws fresh numbers.

A graph is valid for ONE batch shape (rows, row length, number of plan segments, longest segment) - synthetic benches, fixed-length
episodes.  Every `step()` is exactly ONE update (same update-to-data ratio and random streams as the eager loop): the first `warmup`
calls run eagerly (allocator and lazily initialised kernels warm up), a shape is recorded the SECOND time it occurs (a shape that
never recurs is not worth two device synchronisations and an activation pool), at most `max_graphs` graphs live at a time in ONE
shared memory pool (they never replay concurrently), the least recently used one is dropped for a newcomer - but an evicted shape must
recur RECAPTURE_HITS times before it is recorded again and at most `max_graphs` recordings happen per CAPTURE_WINDOW updates, so a
workload with more recurring shapes than graphs settles into eager launches for the overflow instead of recording on every update;
any update without a graph runs eagerly from the same static inputs.
Attention layers (cgpt): the token index / cu_seqlens tables of the batch are built by `_prepare` into pinned buffers (copy nodes in the
graph; their sizes and the longest sequence are part of the shape key), and the counter-keyed dropout masks take their offsets from a
per-update host count (0, 4, 8, ... in program order - baked into the kernel nodes) PLUS a device word that a node of the graph
advances (`ops.dropout_offset_base`): every replay draws fresh masks, forward and backward kernels of one replay agree, and an eager
fallback through `step()` draws exactly what a replay would.
`policy_update_per` > 1 (every published launch script of the reference sets 2, gen_tmuxp_mamba_pomdp.py:81): whether the actor step is due
(reference :450-452) is part of the graph key - an update with and an update without the actor step are two recordings that alternate.
Gradient clipping (reference :239-250, 274-287) stays on the device: norm -> coefficient -> the scale word the flat AdamW kernel reads.
Data-parallel groups: the recording is CUT at each gradient exchange (`cut`) - an update is then three graphs replayed back to back with
the two all-reduces issued eagerly between them on the same stream (works with every backend; nothing waits on the host).
Refused at construction (use the eager `train_one_batch`): side-stream overlap (gru), utd != 1, randomised masks / truncation, the
three-phase Q-guard exchange (RESEL_DP_GUARD=allreduce)."""
import os
from collections import OrderedDict

import numpy as np
import torch

from ..hip import ops
from ..models.flash_attention.TransformerFlashAttention import PackedSeqs
from .sac_full_length_rnn_ensembleQ import DeferredLog


class _StaticLog(DeferredLog):
    def __init__(self, keys, buf, event, host_items):
        dict.__init__(self)
        self._keys, self._buf, self._event = keys, buf, event
        self.set_host(host_items)


class GraphedUpdate:
    PLAN_CAPACITY = 16384                             # plan segments the static buffers hold (64 KB of pinned memory per slot)
    RING = 3                                          # pinned staging slots: the host may be this many updates ahead of the device
    SEEN_CAP = 1024                                   # shape keys remembered (variable-length episodes produce many that never recur)
    RECAPTURE_HITS = 8                                # occurrences an EVICTED shape needs before it is recorded again
    CAPTURE_WINDOW = 256                              # at most `max_graphs` recordings per this many updates: more recurring shapes than graphs run eagerly

    def __init__(self, alg, warmup=3, max_graphs=None):
        why = self.refusal(alg)
        if why:
            raise RuntimeError('GraphedUpdate: ' + why)
        self.alg, self.device = alg, alg.device
        self.graphs = OrderedDict()                   # batch shape key -> CUDAGraph, least recently used first
        if max_graphs is None:                        # four batch shapes; with policy_update_per > 1 every shape has two launch sequences
            max_graphs = 4 * (2 if alg.parameter.policy_update_per > 1 else 1)
        self.warmup, self.max_graphs = warmup, max_graphs
        self._eager_left = warmup                     # updates still to run eagerly before anything is recorded
        self._seen = OrderedDict()                    # batch shape key -> occurrences so far (least recently seen first; pruned to SEEN_CAP)
        self._captures = []                           # step indices of the most recent recordings (capture-rate limit)
        self._steps = 0
        self._amax_generation = ops.AMAX_GENERATION[0]
        self._pool = torch.cuda.graph_pool_handle()   # memory pool shared by every recorded graph (and by the segments of one update)
        self.eager_fallbacks = 0
        # data-parallel groups: the process group's watchdog thread polls events while this thread records - harmless, but a capture in the
        # default `global` mode treats such a call from ANOTHER thread as an error and invalidates the recording
        self._capture_mode = 'thread_local' if alg.grad_sync.active else 'global'
        self._rec = None                              # while recording: dict(segs=[(graph, exchange or None), ...], g=<graph being captured>)
        if alg.grad_sync.active:                      # every rank's Q-guard extrema ride behind the gradients: size the buffer before anything is recorded
            alg.values[0].store.ensure_grad_extra(4 + 4 * alg.grad_sync.world)
        E = alg.target_values[0].uni_network.layer_list[-1].num_ensemble
        self.E = E
        self._draw = type(alg)._select_target_ensemble.__get__(alg)         # the trainer's own host draw
        # static subset buffers sized WITHOUT consuming a draw: evaluate the draw on a scratch numpy stream and put the stream back
        st_np = np.random.get_state()
        if hasattr(alg, '_subset_stream'):
            alg._subset_stream()                      # data-parallel ranks create their shared stream on first use: create it BEFORE its state is saved
        rng_own = getattr(alg, '_subset_rng', None)
        st_own = None if rng_own is None else rng_own.get_state()
        first = np.asarray(self._draw(E))
        np.random.set_state(st_np)
        if st_own is not None:
            rng_own.set_state(st_own)
        self.subset_np = np.ascontiguousarray(first, dtype=np.int32)
        self.subset_i32 = torch.from_numpy(self.subset_np.copy()).to(self.device)
        self.subset_i64 = self.subset_i32.long()
        for opt in (alg.optimizer_value, alg.optimizer_policy):
            opt.enable_device_factors()
        oa = alg.optimizer_alpha                                             # torch AdamW over the entropy coefficient: capturable form
        for g in oa.param_groups:
            g['capturable'] = True
        for st in oa.state.values():
            if torch.is_tensor(st.get('step')):
                st['step'] = st['step'].to(self.device)
        self._plan_dev = torch.empty((self.PLAN_CAPACITY, 4), dtype=torch.int32, device=self.device)
        self._log_dev = torch.zeros(64, dtype=torch.float32, device=self.device)    # static target of the graph's log node
        self._ring = [dict(plan=torch.empty((self.PLAN_CAPACITY, 4), dtype=torch.int32, pin_memory=True),
                           sub=torch.empty(self.subset_np.size, dtype=torch.int32, pin_memory=True),
                           bc=torch.empty(4, dtype=torch.float32, pin_memory=True),
                           log=torch.empty(64, dtype=torch.float32, pin_memory=True),
                           evt=torch.cuda.Event(), used=False, handed=None) for _ in range(self.RING)]
        self._turn = 0
        self._slot = self._ring[0]
        # attention layers: static sequence tables (two descriptions per batch: the batch and its one-slot shift) and the dropout base
        self._seq = None
        self._drop_base = None
        self._drop_seed = 0
        if getattr(alg, '_needs_seq_table', False):
            self._drop_base = torch.zeros(1, dtype=torch.int64, device=self.device)
            ops.dropout_offset_base(self._drop_base)
            if not torch.cuda.default_generators:
                torch.cuda.init()
            dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
            self._drop_seed = torch.cuda.default_generators[dev_index].initial_seed() & ((1 << 64) - 1)

    DROP_STRIDE = 1 << 16                             # offsets one update may consume (4 per mask: 16 384 masks)

    def close(self):
        """Detach the process-wide dropout base again (eager trainers of the same process go back to torch's generator offsets)."""
        if self._drop_base is not None:
            ops.dropout_offset_base(None)
            self._drop_base = None

    def _seq_buffers(self, n_idx, n_cu):
        """Static pinned + device buffers of the two sequence descriptions; growing them invalidates every recorded graph."""
        sq = self._seq
        if sq is None or sq['cap_idx'] < n_idx or sq['cap_cu'] < n_cu:
            ci, cc = max(2 * n_idx, 1024), max(2 * n_cu, 64)
            self.graphs.clear()
            torch.cuda.synchronize(self.device)       # queued copies may still read the old blocks
            sq = self._seq = dict(cap_idx=ci, cap_cu=cc,
                                  idx_host=[[torch.empty(ci, dtype=torch.int64, pin_memory=True) for _ in range(2)] for _ in range(self.RING)],
                                  cu_host=[[torch.empty(cc, dtype=torch.int32, pin_memory=True) for _ in range(2)] for _ in range(self.RING)],
                                  idx_dev=[torch.empty(ci, dtype=torch.int64, device=self.device) for _ in range(2)],
                                  cu_dev=[torch.empty(cc, dtype=torch.int32, device=self.device) for _ in range(2)])
        return sq

    @staticmethod
    def _build_seqs(pl):
        """Host half of `_batch_views` for attention layers (runs while the previous replay is still on the GPU)."""
        rows, T, table = pl['nrow'], pl['longest'], pl['table']
        am = np.zeros((rows, T), dtype=np.int32)
        am[:, :table.shape[1]] = table
        tam = np.concatenate((am[:, 1:], np.zeros((rows, 1), dtype=np.int32)), axis=1)
        return [PackedSeqs.build_host(a, T) for a in (am, tam)]

    def _prepare_seqs(self, built):
        """Both descriptions into the pinned buffers (the previous replay has read them); returns their part of the shape key."""
        sq = self._seq_buffers(max(b[0].size for b in built), max(b[1].size for b in built))
        k = self._turn % self.RING
        for i, (idx, cu, mx, tb) in enumerate(built):
            # staged with numpy (a plain memcpy): a torch CPU copy_ of > 32 768 elements opens an intra-op parallel region over all host threads,
            # which on a loaded host took 10-15 ms per call - longer than the replay it is meant to hide behind
            sq['idx_host'][k][i].numpy()[:idx.size] = idx
            sq['cu_host'][k][i].numpy()[:cu.size] = cu
            sq['idx_dev'][i][:idx.size].copy_(sq['idx_host'][k][i][:idx.size], non_blocking=True)      # stream-ordered, outside the graph
            sq['cu_dev'][i][:cu.size].copy_(sq['cu_host'][k][i][:cu.size], non_blocking=True)
        self._seq_now = [(b[0].size, b[1].size, b[2], b[3]) for b in built]
        return tuple(x for b in built for x in (b[0].size, b[1].size, b[2]))

    def packed_seqs(self):
        """Called by the trainer's `_batch_views` while this object drives the update: views of the static device tables (refreshed by
        `_prepare_seqs` before the replay)."""
        sq = self._seq
        return [PackedSeqs.from_static(sq['idx_dev'][i][:n_idx], sq['cu_dev'][i][:n_cu], mx, tb)
                for i, (n_idx, n_cu, mx, tb) in enumerate(self._seq_now)]

    @staticmethod
    def refusal(alg):
        par = alg.parameter
        if alg.device.type != 'cuda':
            return 'needs a GPU'
        if getattr(alg, 'overlap_value_embedding', False):
            # tried in round 4 (twice; the second time without the tensors' record_stream calls): with the refusal lifted the capture of
            # the gru trainer (target pass and prefetched value embeddings on side streams, forked / joined with events, the same side
            # stream forked more than once per update) ends in a segmentation fault inside capture_end (hipStreamEndCapture) on this ROCm
            # build; round 6 gave every fork of an update its own side stream (`_side_streams`): the same fault.  Captured on ONE stream the
            # recurrences (3 us per step, latency-bound) would run back to back: slower than eager.
            return 'side-stream overlap (gru) is not captured'
        if getattr(alg, 'grad_sync', None) is None:
            return 'not a full-trajectory trainer'
        if alg.grad_sync.active and os.environ.get('RESEL_DP_GUARD', 'bucket') == 'allreduce':
            return 'the three-phase Q-guard exchange (RESEL_DP_GUARD=allreduce) issues collectives inside the target computation'
        if par.utd != 1 or par.randomize_mask or par.random_trunc_traj:
            return 'utd != 1 or randomised masks / truncation change the launch sequence from update to update'
        if not (getattr(alg, 'device_replay', False) and alg.replay_buffer.device_supported(randomize_mask=par.randomize_mask)):
            return 'needs the device-resident replay ring'
        return None

    # ---- called by the trainer while `alg._graph is self` -------------------------------------------------------------------
    def gather(self):
        pl = self._plan
        n = pl['seg'].shape[0]
        dev = self.alg.replay_buffer.gather_planned(self.device, self._plan_dev[:n], pl['max_len'], pl['nrow'], pl['longest'])
        return dev, pl['total_size'], pl['table']

    def cut(self, exchange):
        """Called by the trainer where ranks exchange gradients (`_finish_step`).  Recording: the graph being captured ends here, the
        exchange is remembered behind it and a new graph begins (same pool, same stream) - the exchange is NOT issued while recording
        (nothing has run yet).  Otherwise (eager launch through `step()`): issue it now."""
        rec = self._rec
        if rec is None:
            exchange()
            return
        rec['g'].capture_end()
        rec['segs'].append((rec['g'], exchange))
        rec['g'] = torch.cuda.CUDAGraph()
        rec['g'].capture_begin(pool=self._pool, capture_error_mode=self._capture_mode)

    def _record(self):
        """Record one update as a list of (graph, exchange) segments; one segment unless the trainer cuts (data parallel)."""
        torch.cuda.synchronize(self.device)
        torch.cuda.empty_cache()
        side = self.__dict__.setdefault('_capture_stream', torch.cuda.Stream(device=self.device))
        side.wait_stream(torch.cuda.current_stream(self.device))
        rec = self._rec = dict(segs=[], g=torch.cuda.CUDAGraph())
        try:
            with torch.cuda.stream(side):
                rec['g'].capture_begin(pool=self._pool, capture_error_mode=self._capture_mode)
                try:
                    self._body()                      # recorded, not run: the prepared inputs are consumed by the replay that follows
                finally:
                    rec['g'].capture_end()
                rec['segs'].append((rec['g'], None))
        finally:
            self._rec = None
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        return dict(segs=rec['segs'], log_keys=self._log_keys, log_items=self._log_items)

    def log_node(self, keys, packed, host, items):
        self._log_keys = keys
        self._log_dev[:packed.numel()].copy_(packed)                         # a node of the graph; the read-back follows the replay (_finish)
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
        built = self._build_seqs(pl) if self._drop_base is not None else None
        # staging slot of this update: its copies of three updates ago have long run; the log handed out then is read out before its block is reused
        slot = self._slot = self._ring[self._turn % self.RING]
        if slot['handed'] is not None:
            slot['handed'].resolve()
            slot['handed'] = None
        if slot['used']:
            slot['evt'].synchronize()
        if built is not None:
            key = key + self._prepare_seqs(built)
        slot['plan'].numpy()[:n] = pl['seg']
        self._plan_dev[:n].copy_(slot['plan'][:n], non_blocking=True)        # stream-ordered: behind the previous update's gather
        self._plan = pl
        sub = np.ascontiguousarray(np.asarray(self._draw(self.E)), dtype=np.int32)
        slot['sub'].numpy()[...] = sub
        self.subset_i32.copy_(slot['sub'], non_blocking=True)
        self.subset_i64.copy_(self.subset_i32)
        # is the actor step due in this update (reference :450-452 with utd = 1)?  Part of the key: the two launch sequences are two graphs
        actor_due = alg.grad_num % par.policy_update_per == 0 and par.policy_utd > 0
        for i, opt in enumerate((alg.optimizer_value, alg.optimizer_policy)):
            if i == 0 or actor_due:
                opt.prepare_step(slot['bc'][2 * i:2 * i + 2])
        self._turn += 1
        return key + (bool(actor_due),)

    def _body(self):
        alg = self.alg
        alg._graph = self
        alg._select_target_ensemble = lambda E: self.subset_np
        for opt in (alg.optimizer_value, alg.optimizer_policy):
            opt.device_factors_active = True
        if self._drop_base is not None:
            ops.DROP_OVERRIDE = [self._drop_seed, 0]
        try:
            ops.amax_arena_zero(self.device)          # first node: a replay publishes operand magnitudes into the slots / epochs baked into the graph
            if self._drop_base is not None:
                self._drop_base.add_(self.DROP_STRIDE)                       # a node: every replay moves the masks of the whole update on
            alg.train_one_batch()
        finally:
            ops.DROP_OVERRIDE = None
            alg._graph = None
            del alg._select_target_ensemble
            for opt in (alg.optimizer_value, alg.optimizer_policy):
                opt.device_factors_active = False

    def _finish(self):
        n = len(self._log_keys)
        slot = self._slot
        slot['log'][:n].copy_(self._log_dev[:n], non_blocking=True)          # behind the update on the stream
        slot['evt'].record()
        slot['used'] = True
        items = dict(self._log_items)
        pl = self._plan                               # the host entries of THIS update (the captured dict holds the recorded update's)
        items.update(real_batch_size=pl['total_size'], real_batch_traj_num=pl['nrow'],
                     average_traj_len=self.alg.replay_buffer.size / len(self.alg.replay_buffer))
        slot['handed'] = _StaticLog(list(self._log_keys), slot['log'][:n], slot['evt'], items)
        return slot['handed']

    def step(self):
        """Exactly one update: eager while warming up or while the batch shape has no graph, a replay otherwise."""
        if ops.amax_maintenance() or self._amax_generation != ops.AMAX_GENERATION[0]:
            self.graphs.clear()                       # the magnitude epochs started over: recorded kernel nodes carry epochs of the old generation
            self._amax_generation = ops.AMAX_GENERATION[0]
        key = self._prepare()
        g = self.graphs.get(key)
        self._steps += 1
        if g is None:
            seen = self._seen[key] = self._seen.get(key, 0) + 1
            self._seen.move_to_end(key)
            while len(self._seen) > self.SEEN_CAP:
                self._seen.popitem(last=False)
            # a recording costs two device synchronisations and a whole capture pass: with more recurring shapes than `max_graphs` an
            # evict-and-record-again policy would turn every update into capture + replay.  Hence: an evicted shape has to recur
            # RECAPTURE_HITS times before it is recorded again, and no more than `max_graphs` recordings per CAPTURE_WINDOW updates -
            # whatever has no graph runs eagerly (the same update from the same static inputs).
            self._captures = [t for t in self._captures if self._steps - t < self.CAPTURE_WINDOW]
            throttled = len(self.graphs) >= self.max_graphs and len(self._captures) >= self.max_graphs
            if self._eager_left > 0 or seen < 2 or throttled:      # warm-up updates / a shape on its first visit: the same update, launched eagerly
                self._eager_left = max(0, self._eager_left - 1)
                self.eager_fallbacks += 1
                self._body()
                return self._finish()
            if len(self.graphs) >= self.max_graphs:
                old_key, _ = self.graphs.popitem(last=False)       # least recently used; its activations return to the shared pool
                self._seen[old_key] = 2 - self.RECAPTURE_HITS
            self._captures.append(self._steps)
            g = self.graphs[key] = self._record()
        self.graphs.move_to_end(key)
        self._log_keys, self._log_items = g['log_keys'], g['log_items']      # of THIS launch sequence (with / without the actor step)
        for seg, exchange in g['segs']:
            seg.replay()
            if exchange is not None:
                exchange()                            # the gradient all-reduce, eagerly between two replays on the same stream
        if ops.AMAX_VERIFY:                           # the recorded update contains the check kernels: read their verdict behind the replay
            ops.amax_verify_raise(self.device)
        return self._finish()

    @property
    def graph(self):                                  # the most recently recorded graph (tests): its first segment
        return next(reversed(self.graphs.values()))['segs'][0][0] if self.graphs else None
