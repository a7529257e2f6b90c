"""Full-trajectory recurrent SAC with an ensemble critic: `train_one_batch()` - THE hot path
(reference offpolicy_rnn/algorithm/sac_full_length_rnn_ensembleQ.py:297-467).

Same contract (no arguments, returns the log dict with the reference's keys, `grad_num` gates the actor step) and the
same arithmetic per update:
    sample packed trajectories -> [no grad] actor + target critics on (s', s, a) -> REDQ / min target with the
    Q-guard clamp -> critic step -> soft target update -> actor step (+ alpha step) on the live critics.
What is organised differently for the MI355X:
  * the sampled batch crosses PCIe ONCE as a single fp32 array (all fields, validity and the derived reset /
    validity flags are columns of it); field tensors are column views on the device;
  * target, Q-guard state and the running extrema stay on the device (`ops.sac_target`);
  * losses are back-propagated as masked SUMS; the 1 / valid_num normalisation is applied inside the flat AdamW
    kernel from a device scalar that rides in the gradient buffer - which is what makes the data-parallel version a
    single all-reduce (parallel/data_parallel.py) with a global normalisation identical to the reference's;
  * the actor step back-propagates only into the actor (`backward(inputs=...)`), skipping the critic weight gradients
    that the reference computes and throws away;
  * every scalar of the log dict is gathered on the device and fetched with one transfer at the end.
"""
from typing import Dict

import os

import numpy as np
import torch

from ..buffers.transition_buffer.nested_replay_memory import NestedMemoryArray as NestedTransitionMemoryArray
from ..hip import ops
from ..models.flash_attention.TransformerFlashAttention import PackedSeqs
from ..parallel.data_parallel import GradSync
from ..policy_value_models.make_models import make_policy_model
from ..utility.pinned import PinnedRing

FUSED_LOSSES = FUSED_Q = FUSED_A = True      # the masked losses as one forward + one backward kernel each (csrc/losses.hip); CPU tensors keep the torch spelling
from ..utility.q_value_guard import QValueGuard
from .sac import SAC

FIELDS = ('state', 'last_state', 'action', 'last_action', 'next_state', 'done', 'mask', 'reward', 'reward_input', 'timeout', 'start')


import contextlib


@contextlib.contextmanager
def _frozen_parameters(model):
    params = [p for p in model.parameters() if p.requires_grad]
    for p in params:
        p.requires_grad_(False)
    try:
        yield
    finally:
        for p in params:
            p.requires_grad_(True)


@contextlib.contextmanager
def _ensemble_subset(model, member_index):
    from ..models.ensemble_linear_model import EnsembleLinear
    layers = [l for l in model.uni_network.layer_list if isinstance(l, EnsembleLinear)]
    for l in layers:
        l.member_index = member_index
    try:
        yield
    finally:
        for l in layers:
            l.member_index = None


class DeferredLog(dict):
    """The update's log dict.  Host-side entries are plain items; the device scalars travel in one asynchronous copy into
    pinned memory and become floats on first access (`resolve()`), so a caller that does not read them - a training loop
    logging every n-th update, bench.py - never stalls the launch queue on the update it has just enqueued."""

    def __init__(self, keys, packed, pinned):
        super().__init__()
        self._keys = keys
        if pinned:
            self._buf = torch.empty(packed.shape, dtype=packed.dtype, pin_memory=True)
            self._buf.copy_(packed, non_blocking=True)
            self._event = torch.cuda.Event()
            self._event.record()
        else:
            self._buf, self._event = packed.detach().cpu(), None

    def set_host(self, items):
        dict.update(self, items)

    def resolve(self):
        if self._keys is not None:
            if self._event is not None:
                self._event.synchronize()
            vals = dict(zip(self._keys, self._buf.tolist()))
            if 'actor_loss' in vals:
                vals['actor_loss'] = (vals['actor_loss'],)      # a 1-tuple upstream (reference :430); kept
            self._keys = None
            pending = dict(self)                                # host entries override / follow, as in the reference order
            dict.clear(self)
            dict.update(self, vals)
            dict.update(self, pending)
        return self

    def __getitem__(self, k):
        if not dict.__contains__(self, k):
            self.resolve()
        return dict.__getitem__(self, k)

    def get(self, k, default=None):
        if not dict.__contains__(self, k):
            self.resolve()
        return dict.get(self, k, default)

    def __contains__(self, k):
        return dict.__contains__(self, k) or dict.__contains__(self.resolve(), k)

    def __iter__(self):
        return dict.__iter__(self.resolve())

    def __len__(self):
        return dict.__len__(self.resolve())

    def keys(self):
        return dict.keys(self.resolve())

    def items(self):
        return dict.items(self.resolve())

    def values(self):
        return dict.values(self.resolve())

    def __repr__(self):
        return dict.__repr__(self.resolve())


class SACFullLengthRNNEnsembleQ(SAC):
    def __init__(self, parameter):
        super().__init__(parameter)
        assert self.parameter.value_net_num == 1
        for item in self.parameter.value_layer_type:
            assert item.startswith('e'), 'the critic MLP must be an ensemble (efc-<E>) stack'
        for net in (self.values[0].embedding_network.layer_list + self.target_values[0].embedding_network.layer_list
                    + self.values[0].uni_network.layer_list + self.target_values[0].uni_network.layer_list):
            if hasattr(net, 'desire_ndim'):
                net.desire_ndim = 4
            if hasattr(net, 'in_proj') and hasattr(net.in_proj, 'desire_ndim'):
                net.in_proj.desire_ndim = 4
        self.logger(f'replay buffer skip len: {self._get_skip_len()}')
        self.amp_scalar = self.amp_scalar_critic = None            # fp32 / bf16 paths need no loss scaling
        self.replay_buffer = NestedTransitionMemoryArray(self.parameter.max_buffer_transition_num, self.env_info['max_trajectory_len'],
                                                         additional_history_len=self._get_skip_len())
        self.Q_guard = QValueGuard(guard_min=True, guard_max=True, decay_ratio=1.0 if self.discrete_env else 1 - 1e-3,
                                   device=self.device)
        # kept for API parity: built, hard-copied, never updated; only the non-REDQ TD3 trainer reads it
        self.target_policy = make_policy_model(self.policy_args, self.base_algorithm, self.discrete_env)
        self.target_policy.to(self.device)
        self.target_policy.copy_weight_from(self.policy, tau=0.0)
        self.target_policy.eval()
        self.grad_sync = GradSync()
        self._graph = None                             # set while a captured update is being recorded / replayed (graphed_update.py)
        self._subset_rng = None                        # REDQ subset stream: None = numpy's global stream (see _subset_stream)
        self._pinned = PinnedRing(torch.float32)       # staging blocks of the host-built batch (one event per block)
        self._needs_seq_table = any(lid.startswith('cgpt') for net in (self.values[0].uni_network, self.values[0].embedding_network,
                                                                       self.policy.uni_network, self.policy.embedding_network)
                                    for lid in net.layer_type)
        self._stats = torch.zeros(2, dtype=torch.float32, device=self.device)
        # latency-bound value embeddings (gru) of the two graph-free critic passes run on a side stream next to the policy pass
        self.overlap_value_embedding = (self.device.type == 'cuda' and os.environ.get('RESEL_OVERLAP_EMBEDDING', '1') != '0'
                                        and not self.discrete_env and any(lid == 'gru' for lid in self.values[0].embedding_network.layer_type))
        self._side_stream = self._target_stream = None
        self._side_streams, self._fork_idx = [], 0
        self._shared_policy_out = None
        self._share_this_update = False
        self.share_policy_pass = self._policy_pass_shareable()
        if self.share_policy_pass:              # the shared pass records an autograd graph inside the target computation: keep that on one stream
            self.overlap_value_embedding = False

    step = property(lambda self: self.train_one_batch)          # north_star's "algorithm.step()" alias
    device_replay = True        # keep a device mirror of the replay ring and assemble sampled batches on the GPU (CUDA only)

    def _get_skip_len(self):
        skip = 0
        for net in (self.values[0].uni_network, self.values[0].embedding_network, self.policy.uni_network, self.policy.embedding_network):
            for i, lid in enumerate(net.layer_type):
                if 'smamba' in lid:
                    skip = max(net.layer_list[i].d_conv, skip)
                elif 'mamba' in lid:
                    skip = max(net.layer_list[i].mixer.d_conv, skip)
                elif 'conv1d' in lid:
                    skip = max(net.layer_list[i].d_conv, skip)
        return skip + 1

    def _get_whether_require_amp(self):
        return False

    # state cleared / masked at the first token of a trajectory; cgpt attends inside per-trajectory segments whose table
    # is shifted with the tokens (`target_attention_mask`, reference :358-366)
    _SHIFT_INVARIANT_IDS = ('smamba', 'mamba', 'gilr', 'lru', 'conv1d', 'cgpt')

    def _policy_pass_shareable(self) -> bool:
        """The target pass evaluates the policy on (s', s, a): the SAME token sequence as the actor pass on (s, s_prev,
        a_prev), one slot earlier (that is what the pre-step slot and the `total_*` flags of the reference build,
        sac_full_length_rnn_ensembleQ.py:338-378), with the SAME parameters (the actor is updated after both).  When every
        layer's output at a token depends only on the tokens of its own trajectory - pointwise layers and recurrent layers
        that reset / mask at the trajectory start - the actor pass is the target pass shifted by one slot, so ONE policy
        forward (with a graph) serves both and the second is skipped.  Not for `gru` (no reset: the leading padding slots
        differ between the two passes), layers with active dropout (`cgpt_*_p0.1`), the TD3 trainer that smooths a frozen target policy, discrete heads,
        RESEL_SHARE_POLICY_PASS=0."""
        if os.environ.get('RESEL_SHARE_POLICY_PASS', '1') == '0' or self.discrete_env or not self.target_from_live_policy:
            return False
        if self.parameter.randomize_first_hidden:        # the two passes would start from two different random states
            return False
        if type(self)._next_action is not SACFullLengthRNNEnsembleQ._next_action:
            return False
        if self.base_algorithm not in ('sac', 'td3'):
            return False
        for net in (self.policy.embedding_network, self.policy.uni_network):
            for lid in net.layer_type:
                if not (lid == 'fc' or lid.startswith('efc') or lid.startswith(self._SHIFT_INVARIANT_IDS)):
                    return False
        return not any(isinstance(m, torch.nn.Dropout) and m.p > 0 for mod in self.policy.contextual_modules.values() for m in mod.modules())

    def _mask_mean(self, data: torch.Tensor, mask: torch.Tensor, valid_num) -> torch.Tensor:
        return (data * mask).sum() / valid_num

    # ------------------------------------------------------------------------------------------ batch
    def _upload_batch(self, batch, valid, table):
        """Host fix-ups + ONE host->device copy.  Returns dict of device views."""
        arr = self.replay_buffer._last_batch_array                   # (rows, T', W) fp32: every field is a column range
        rows, T, W = arr.shape
        R = self.replay_buffer.name2range
        need = rows * T * (W + 3)
        staged = self._pinned.stage(need, self.device, rows * self.replay_buffer.max_traj_step * (W + 3)).view(rows, T, W + 3)
        host = staged.numpy()
        host[..., :W] = arr
        start = arr[..., R['start'][0]]
        v = valid[..., 0]
        # flag surgery (reference :338-342): the target pass sees the slot before each sequence as valid / not-start
        tv = v.copy()
        tv[:, :-1][np.diff(v, axis=1) == 1] = 1
        ts = start.copy()
        ts[:, :-1][np.diff(start, axis=1) == -1] = 0
        host[..., W], host[..., W + 1], host[..., W + 2] = v, tv, ts
        d0, t0 = R['done'][0], R['timeout'][0]
        host[..., d0][arr[..., t0] > 0] = 0                          # time-limit terminations bootstrap
        dev = self._pinned.upload(staged, self.device)
        return self._batch_views(dev, table)

    def _batch_views(self, dev, table):
        """Field / flag views of the device batch array (rows, T', W + 3)."""
        rows, T = dev.shape[:2]
        W = dev.shape[2] - 3
        R = self.replay_buffer.name2range
        out = {name: dev[..., R[name][0]:R[name][1]] for name in FIELDS}
        out['valid'], out['total_valid'], out['total_start'] = dev[..., W:W + 1], dev[..., W + 1:W + 2], dev[..., W + 2:W + 3]
        # the per-token scalars (flags, reward, done ...) as ONE planar [k, rows, T] copy: every kernel that takes them wants a dense
        # [rows * T] vector, and as column views of the batch array (stride W + 3) each use made its own contiguous copy - 24 small
        # launches per update
        ones = [n for n in out if out[n].shape[-1] == 1]
        if ones and dev.is_cuda:
            key = (tuple(ones), dev.device)
            idx = self.__dict__.setdefault('_planar_idx', {}).get(key)
            if idx is None:
                cols = [R[n][0] if n in R else {'valid': W, 'total_valid': W + 1, 'total_start': W + 2}[n] for n in ones]
                idx = self._planar_idx[key] = torch.tensor(cols, dtype=torch.int64, device=dev.device)
            planar = dev.view(rows * T, W + 3).index_select(1, idx).t().contiguous()          # [k, rows * T]
            for i, n in enumerate(ones):
                out[n] = planar[i].view(rows, T, 1)
        # per-row sequence-length tables (reference :358-366) are consumed by attention layers only
        out['attention_mask'] = out['target_attention_mask'] = None
        if self._needs_seq_table and self._graph is not None:
            # captured update: the tables were built by GraphedUpdate._prepare into pinned buffers; copy nodes + static device views
            out['attention_mask'], out['target_attention_mask'] = self._graph.packed_seqs()
        elif self._needs_seq_table:
            am = np.zeros((rows, T), dtype=np.int32)
            am[:, :table.shape[1]] = table
            tam = np.concatenate((am[:, 1:], np.zeros((rows, 1), dtype=np.int32)), axis=1)
            # built on the host (the table is host data anyway): token indices + cu_seqlens for the var-len attention kernel
            out['attention_mask'], out['target_attention_mask'] = PackedSeqs(am, T, self.device), PackedSeqs(tam, T, self.device)
        return out

    def _make_hidden(self, model, rows, start, mask, attn):
        if self.parameter.randomize_first_hidden:
            h = model.make_rnd_init_state(rows, device=self.device)
        else:
            h = model.make_init_state(rows, device=self.device)
        h.set_rnn_start(start)
        h.set_mask(mask)
        h.set_attention_concat_mask(attn)
        return h

    # ------------------------------------------------------------------------------------------ target
    def _prefetch_value_embedding(self, model, args, hidden):
        if not self.overlap_value_embedding:
            return
        # one side stream PER FORK of an update (two: the target critic's and the actor step's embedding): a stream that has been forked,
        # joined and is forked again inside one hipGraph capture ends hipStreamEndCapture with a segmentation fault on this ROCm build
        k, self._fork_idx = self._fork_idx, self._fork_idx + 1
        while len(self._side_streams) <= k:
            self._side_streams.append(torch.cuda.Stream(device=self.device))
        self._side_stream = self._side_streams[0]
        model.prefetch_embedding(args, hidden, self._side_streams[k])

    target_from_live_policy = True        # (the non-REDQ TD3 trainer evaluates its frozen target policy instead)

    def _next_action(self, b, hidden):
        """(a', log pi(a'|s')) on the shifted inputs (s', s, a)."""
        net = self.policy if self.target_from_live_policy else self.target_policy
        mean, _, sample, logp, _, _ = net.forward(b['next_state'], b['state'], b['action'], hidden, b['reward'])
        return self._target_action(mean, sample, logp)

    def _target_action(self, mean, sample, logp):
        """What the target uses from the policy head's (mean, sample, log-prob); TD3 trainers smooth the mean instead."""
        return sample, logp

    def _subset_on_device(self, subset: np.ndarray, num_ensemble: int = 0, as_long: bool = False) -> torch.Tensor:
        """int32 device copy of a critic-subset index vector without a per-update host->device copy (a pageable copy blocks
        the host until the launch queue has drained): all ordered subsets of that size are uploaded ONCE as a table
        (8 critics, REDQ pairs: 56 rows) and a row view is returned; oversized tables fall back to a per-subset cache."""
        import itertools
        import math
        if self._graph is not None and subset is self._graph.subset_np:   # captured update: the drawn subset sits in static buffers
            return self._graph.subset_i64 if as_long else self._graph.subset_i32
        sub = np.ascontiguousarray(subset, dtype=np.int32)
        m, E = int(sub.size), int(max(num_ensemble, sub.max() + 1))
        tables = self.__dict__.setdefault('_subset_tables', {})
        if (E, m) not in tables:
            rows = None
            if math.perm(E, m) <= 4096:
                perms = list(itertools.permutations(range(E), m))
                t32 = torch.tensor(perms, dtype=torch.int32).to(self.device)
                rows = ({p: i for i, p in enumerate(perms)}, t32, t32.long())
            tables[(E, m)] = rows
        rows = tables[(E, m)]
        if rows is not None and tuple(sub.tolist()) in rows[0]:
            return rows[2 if as_long else 1][rows[0][tuple(sub.tolist())]]
        cache = self.__dict__.setdefault('_subset_cache', {})
        key = (sub.tobytes(), as_long)
        if key not in cache:
            t = torch.from_numpy(sub.copy()).to(self.device)
            cache[key] = t.long() if as_long else t
        return cache[key]

    def _select_target_ensemble(self, num_ensemble: int) -> np.ndarray:
        return np.arange(num_ensemble)                               # plain ensemble-min (REDQ trainers override)

    def _subset_stream(self):
        """numpy stream of the REDQ subset draws.  One process: the global stream, draw for draw as the reference.  Data parallel:
        a dedicated stream seeded identically on every rank (`parameter.seed`), so that all ranks take the minimum over the
        SAME critics while each samples its own rows from its own global stream - the update then equals the single-process
        update over the global batch (tests/test_data_parallel*.py)."""
        if self.grad_sync.active and self._subset_rng is None:
            self._subset_rng = np.random.RandomState(int(self.parameter.seed) + 7919)
        return self._subset_rng if self._subset_rng is not None else np.random

    def _guard_exchange(self):
        """Data parallel: the Q-guard sees the extrema of the GLOBAL batch.  Default: no collective of its own - the rank-local
        extrema ride in the critic's gradient bucket (`_finish_step`) and reach the guard behind that all-reduce; the guard of the
        reference acts on the NEXT update's target only (utility/q_value_guard.py:22-38), so this is the single-process guard, not
        an approximation.  RESEL_DP_GUARD=allreduce keeps the three-phase form with two 2-float MAX all-reduces inside the target."""
        if not self.grad_sync.active:
            return {}
        if os.environ.get('RESEL_DP_GUARD', 'bucket') == 'allreduce':
            return dict(reduce_max=self.grad_sync.all_reduce_max_)
        if getattr(self, '_guard_ext', None) is None or self._guard_ext.device != self.Q_guard.state.device:
            self._guard_ext = torch.zeros(4, dtype=torch.float32, device=self.Q_guard.state.device)
        self._guard_ext_fresh = True
        return dict(local_ext=self._guard_ext)

    def _target_Q_discrete(self, b, policy_hidden, target_hiddens, stats):
        """Discrete-action target (reference sac_full_length_rnn_redq.py:52-72): V(s') = sum_a pi(a|s') (min_subset Q'(s', a) -
        alpha log pi(a|s')); the last action enters the networks as a one-hot vector.  Guard clamp / update, done masking
        and the batch statistics go through the same fused kernel as the continuous target (one pseudo-critic, no log-prob)."""
        with torch.no_grad():
            onehot = self.policy.action2onehot(b['action'])
            _, _, sample, logp, _, _ = self.policy.forward(b['next_state'], b['state'], onehot, policy_hidden, b['reward'])
            tv = self.target_values[0]
            E = tv.uni_network.layer_list[-1].num_ensemble
            subset = np.asarray(self._select_target_ensemble(E))
            if subset.size < E:
                with _ensemble_subset(tv, self._subset_on_device(subset, E, as_long=True)):
                    q = tv.forward(b['next_state'], b['state'], onehot, sample, target_hiddens[0], b['reward'])[0]
            else:
                q = tv.forward(b['next_state'], b['state'], onehot, sample, target_hiddens[0], b['reward'])[0][self._subset_on_device(subset, E, as_long=True)]
            alpha = self.log_sac_alpha.detach().exp()
            v = ((q.min(dim=0).values - alpha * logp) * logp.exp()).sum(dim=-1, keepdim=True)
            return ops.sac_target(v.unsqueeze(0).contiguous(), self._subset_on_device(np.arange(1), 1), None, self.log_sac_alpha.detach(),
                                  b['reward'], b['done'], b['mask'], self.parameter.gamma, self.Q_guard.state, stats, **self._guard_exchange())

    def get_target_Q(self, b, policy_hidden, target_hiddens, stats):
        if self.discrete_env:
            return self._target_Q_discrete(b, policy_hidden, target_hiddens, stats)
        self._shared_policy_out = None
        if self._share_this_update:
            with torch.enable_grad():           # one policy forward with a graph: the actor step reuses it one slot later
                emb_in = self.policy.get_embedding_input(b['next_state'], b['state'], b['action'], b['reward'])
                self._shared_policy_out = self.policy.meta_forward(emb_in, b['next_state'], policy_hidden, False)[0]
        self._prefetch_value_embedding(self.target_values[0], (b['next_state'], b['state'], b['action'], b['reward']), target_hiddens[0])
        with torch.no_grad():
            if self._shared_policy_out is not None:     # same head, same random draws as policy.forward would make
                sample, logp = self._target_action(*self.policy.process_model_out(self._shared_policy_out.detach()))
            else:
                sample, logp = self._next_action(b, policy_hidden)
            tv = self.target_values[0]
            E = tv.uni_network.layer_list[-1].num_ensemble
            subset = np.asarray(self._select_target_ensemble(E))
            if subset.size < E:
                # REDQ: only the sampled target critics are evaluated (same numbers: the others were never used)
                with _ensemble_subset(tv, self._subset_on_device(subset, E, as_long=True)):
                    q = tv.forward(b['next_state'], b['state'], b['action'], sample, target_hiddens[0], b['reward'])[0]
                idx = self._subset_on_device(np.arange(subset.size), subset.size)
            else:
                q = tv.forward(b['next_state'], b['state'], b['action'], sample, target_hiddens[0], b['reward'])[0]
                idx = self._subset_on_device(subset, E)
            return ops.sac_target(q, idx, logp if self.base_algorithm == 'sac' else None, self.log_sac_alpha.detach(), b['reward'],
                                  b['done'], b['mask'], self.parameter.gamma, self.Q_guard.state, stats, **self._guard_exchange())

    # ------------------------------------------------------------------------------------------ losses
    actor_q_reduce = 'min'                           # how the actor objective reduces the critics (REDQ trainers: 'mean'); names the fused kernel's mode

    def _q_for_policy(self, qs: torch.Tensor) -> torch.Tensor:
        return qs.min(dim=0).values                                  # (REDQ trainers use the ensemble mean)

    def _actor_objective(self, alpha, logp, q_pi):
        return alpha * logp - q_pi                                   # (TD3 trainers drop the entropy term)

    def _clip(self, store, model, max_norm, emb_max, scale):
        """Optional gradient clipping on the *normalised* gradient (reference :239-250, 274-287).  Nothing here reads a device value on
        the host: the norm stays a device scalar (it joins the log's packed scalars), so a clipped update is capturable."""
        gnorm = 0.0
        if max_norm is None and emb_max is None:
            return gnorm
        if emb_max is None and store.grad.is_cuda:
            # norm clipping alone rewrites nothing: ||g / count|| = scale * sqrt(sum g^2) from ONE read of the flat buffer, and the
            # coefficient min(1, max_norm / (norm + 1e-6)) (torch.nn.utils.clip_grad_norm_) is folded into the scale word AdamW reads
            gnorm = ops.sumsq(store.grad[:store.numel]).sqrt_().mul_(scale)
            scale.mul_((max_norm / (gnorm + 1e-6)).clamp_(max=1.0))
            return gnorm[0]
        store.grad[:store.numel].mul_(scale)
        scale.fill_(1.0)
        if max_norm is not None:
            gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm, norm_type=2)
        if emb_max is not None:
            torch.nn.utils.clip_grad_value_(model.embedding_network.parameters(), emb_max)
            for layer in model.embedding_network.layer_list:
                for sub in getattr(layer, 'layers', []):
                    if hasattr(sub, 'mixer'):
                        torch.nn.utils.clip_grad_value_([sub.mixer.A_log], 1e-3)
            gnorm = 0.0
        return gnorm

    def _finish_step(self, optimizer, store, local_count, overlap=None, guard=False):
        """Exchange (sum) the flat gradient + the local valid count, then AdamW with grad / global count.  Data parallel: the
        all-reduce is issued on the exchange stream as soon as the last gradient is in the flat buffer; `overlap()` (work
        that does not touch the gradient buffer: the step's log scalars) runs on the compute stream meanwhile, which
        joins before anything reads the reduced buffer."""
        store.collect_grads()
        gs = self.grad_sync
        guard_slots = gs.active and guard and getattr(self, '_guard_ext_fresh', False)
        if guard_slots:                                # this rank's row of the [world][4] extrema block behind the four standing slots
            store.ensure_grad_extra(4 + 4 * gs.world)
            o = store.numel + 4 + 4 * gs.rank
            store.grad[o:o + 4] = self._guard_ext
        store.grad[store.numel] = local_count
        if gs.active and self._graph is not None:
            # captured update: the recording is CUT here - the exchange runs eagerly between two graph replays (any backend, gloo included)
            if overlap is not None:
                overlap()
            grad = store.grad
            self._graph.cut(lambda: gs.all_reduce_(grad))
        else:
            gs.all_reduce_async_(store.grad)
            if overlap is not None:
                overlap()
            gs.wait()
        if guard_slots:                                # every rank's extrema are here: initialise / update the guard from the global batch
            ops.guard_apply_slots(store.grad[store.numel + 4:store.numel + 4 + 4 * gs.world], gs.world, self.Q_guard.state)
            self._guard_ext_fresh = False
        return 1.0 / store.grad[store.numel:store.numel + 1]

    # ------------------------------------------------------------------------------------------ the update
    def train_one_batch(self) -> Dict:
        from ..models.rnn_base import training_pass
        with training_pass():                                    # no pass of an update reads the per-layer output sequences (rnn_base.py)
            return self._train_one_batch()

    def _train_one_batch(self) -> Dict:
        par = self.parameter
        self._fork_idx = 0
        if self._graph is None and self.device.type == 'cuda':
            ops.amax_maintenance()                               # update boundary: the magnitude epochs may start over here (hip/ops.py)
        self.policy.to(self.device)
        self.optimizer_policy.to(self.device)
        policy_update_cnt = 0
        scal: Dict[str, torch.Tensor] = {}
        host: Dict[str, float] = {}
        for utd_idx in range(par.utd):
            self.timer.register_point(tag='sample_trajs', level=2)
            if self._graph is not None:
                # captured update (algorithm/graphed_update.py): the host half of the sampling ran before the replay, the plan sits in
                # static buffers; here only the device half is enqueued (a memcpy node + the gather kernel)
                dev, batch_size, table = self._graph.gather()
                self.timer.register_end(level=2)
                b = self._batch_views(dev, table)
            elif getattr(self, 'device_replay', False) and self.device.type == 'cuda' \
                    and self.replay_buffer.device_supported(randomize_mask=par.randomize_mask):
                # device-resident ring: the host decides WHICH trajectories go where, the batch array is built on the GPU
                dev, batch_size, table = self.replay_buffer.sample_trajs_device(
                    self.device, par.sac_batch_size, None, random_trunc_traj=par.random_trunc_traj, nest_stack_trajs=self.allow_nest_stack)
                self.timer.register_end(level=2)
                b = self._batch_views(dev, table)
            else:
                batch, batch_size, valid, table = self.replay_buffer.sample_trajs(
                    par.sac_batch_size, None, randomize_mask=par.randomize_mask, valid_number_post_randomized=par.valid_number_post_randomized,
                    equalize_data_of_each_traj=True, random_trunc_traj=par.random_trunc_traj, nest_stack_trajs=self.allow_nest_stack)
                self.timer.register_end(level=2)
                b = self._upload_batch(batch, valid, table)
            rows = b['state'].shape[0]
            value, target_value = self.values[0], self.target_values[0]
            # hidden states + side channels: the target pass uses the one-slot-earlier flags (reference :368-378)
            target_policy_hidden = self._make_hidden(self.policy, rows, b['total_start'], b['total_valid'], b['target_attention_mask'])
            target_hiddens = [self._make_hidden(target_value, rows, b['total_start'], b['total_valid'], b['target_attention_mask'])]
            value_hiddens = [self._make_hidden(value, rows, b['start'], b['valid'], b['attention_mask'])]
            policy_hidden = self._make_hidden(self.policy, rows, b['start'], b['valid'], b['attention_mask'])
            alpha_detach = self.log_sac_alpha.exp().detach()
            mask = b['mask']

            # 1. target (no grad); guard clamp/update + max|target| + sum(mask) happen inside the fused kernel
            actor_due = self.grad_num % par.policy_update_per == 0 and (utd_idx + 1) / par.utd * par.policy_utd > policy_update_cnt
            self._share_this_update = self.share_policy_pass and actor_due
            self.policy.eval()
            target_done = None
            if self.overlap_value_embedding:
                # latency-bound layers: the whole (graph-free) target computation goes to a second stream, so that the critic's
                # forward below (main stream) runs beside the target policy pass and the target critic's embedding pass
                main = torch.cuda.current_stream(self.device)
                if self._target_stream is None:
                    self._target_stream = torch.cuda.Stream(device=self.device)
                self._target_stream.wait_stream(main)
                with torch.cuda.stream(self._target_stream):
                    target_Q = self.get_target_Q(b, target_policy_hidden, target_hiddens, self._stats)
                    target_done = torch.cuda.Event()
                    target_done.record(self._target_stream)
            else:
                target_Q = self.get_target_Q(b, target_policy_hidden, target_hiddens, self._stats)
            valid_num = self._stats[1]

            # 2. critic step
            value.train()
            q = value.forward(b['state'], b['last_state'], b['last_action'], b['action'], value_hiddens[0], b['reward_input'])[0]
            if target_done is not None:
                main.wait_event(target_done)
                target_Q.record_stream(main)
            if self.discrete_env:                                   # Q of the action taken (reference :158)
                q = q.gather(-1, b['action'].long().unsqueeze(0).expand(q.shape[0], -1, -1, -1))
            if FUSED_Q and q.is_cuda and q.dtype == torch.float32:
                q_loss_sum = ops.masked_q_loss(q, target_Q, mask)         # one forward + one backward kernel (reference :105-114, :80-81)
            else:
                q_loss_sum = ((q - target_Q.unsqueeze(0)).pow(2).sum(dim=0) * mask).sum()
            self.optimizer_value.zero_grad()
            q_loss_sum.backward()
            scale = self._finish_step(self.optimizer_value, value.store, valid_num, guard=True,
                                      overlap=lambda: scal.update(critic_loss=q_loss_sum.detach() / valid_num))
            q_grad_norm = self._clip(value.store, value, par.value_max_gradnorm, par.value_embedding_max_gradnorm, scale)
            self.optimizer_value.step(grad_scale=scale)
            host['value_grad_norm'] = q_grad_norm

            # 3. soft target update: one kernel over the flat buffers
            self._value_update(tau=par.sac_tau)
            value.eval()
            self.policy.train()

            # 4. actor (+ alpha) step
            if self.grad_num % par.policy_update_per == 0 and (utd_idx + 1) / par.utd * par.policy_utd > policy_update_cnt:
                self._prefetch_value_embedding(value, (b['state'], b['last_state'], b['last_action'], b['reward_input']), value_hiddens[0])
                if self._shared_policy_out is not None:            # the target pass's head outputs, one slot later
                    out2 = torch.nn.functional.pad(self._shared_policy_out[:, :-1], (0, 0, 1, 0))
                    self._shared_policy_out = None
                    action_mean, act_sample, log_prob = self.policy.process_model_out(out2)
                else:
                    action_mean, _, act_sample, log_prob, _, _ = self.policy.forward(b['state'], b['last_state'], b['last_action'],
                                                                                    policy_hidden, b['reward_input'])
                act_in = act_sample if self.base_algorithm == 'sac' else action_mean
                # the actor objective differentiates Q only w.r.t. the action: the critic's parameters are frozen while its graph
                # is recorded, so that the backward does not form the critic weight gradients the reference computes and drops
                with _frozen_parameters(value):
                    q_pi = value.forward(b['state'], b['last_state'], b['last_action'], act_in, value_hiddens[0], b['reward_input'],
                                         detach_embedding=True)[0]
                if self.discrete_env:                               # expectation over the action distribution (reference redq :85-86)
                    objective = (self._actor_objective(alpha_detach, log_prob, self._q_for_policy(q_pi)) * log_prob.exp()).sum(dim=-1, keepdim=True)
                    log_prob = (log_prob * log_prob.exp()).sum(dim=-1, keepdim=True)          # logged as -entropy (:425)
                    actor_sum = (objective * mask).sum()
                    lp_sum = None
                elif FUSED_A and q_pi.is_cuda and q_pi.dtype == torch.float32 and self.base_algorithm == 'sac' and self.actor_q_reduce in ('min', 'mean'):
                    # sum mask (alpha logp - red_e Q) and sum mask logp from one pass, gradients from one more (reference redq :37-49)
                    actor_sum, lp_sum = ops.masked_actor_loss(log_prob, q_pi, mask, self.log_sac_alpha, True, self.actor_q_reduce == 'min')
                else:
                    objective = self._actor_objective(alpha_detach, log_prob, self._q_for_policy(q_pi))
                    actor_sum = (objective * mask).sum()
                    lp_sum = None
                if lp_sum is None:
                    lp_sum = (log_prob.detach() * mask).sum()
                self.optimizer_policy.zero_grad()
                actor_sum.backward(inputs=self.policy.parameters())
                pstore = self.policy.store
                tune_alpha = not par.no_alpha_auto_tune
                if tune_alpha:     # d/d log_alpha of -sum(mask * log_alpha * (logp + H)) rides in the second spare slot
                    pstore.grad[pstore.numel + 1] = -(lp_sum + self.target_entropy * valid_num)
                scale = self._finish_step(self.optimizer_policy, pstore, valid_num, overlap=lambda: scal.update(
                    log_prob=lp_sum / valid_num, actor_loss=actor_sum.detach() / valid_num))
                pi_grad_norm = self._clip(pstore, self.policy, par.policy_max_gradnorm, par.policy_embedding_max_gradnorm, scale)
                self.optimizer_policy.step(grad_scale=scale)
                if tune_alpha:
                    g_alpha = pstore.grad[pstore.numel + 1:pstore.numel + 2] / pstore.grad[pstore.numel:pstore.numel + 1]
                    scal['alpha_loss'] = (self.log_sac_alpha.detach() * g_alpha)[0]
                    self.log_sac_alpha.grad = g_alpha.clone()
                    self.optimizer_alpha.step()
                    with torch.no_grad():
                        self.log_sac_alpha.clamp_max_(1)
                scal['policy_l2_norm_square'] = self.policy.l2_norm_square()
                host['policy_grad_norm'] = pi_grad_norm
                policy_update_cnt += 1
        self.policy.to(self.sample_device)
        scal.update(log_alpha=self.log_sac_alpha.detach()[0], target_q_max=self._stats[0], clip_min=self.Q_guard.state[0],
                    clip_max=self.Q_guard.state[1], q1_l2_norm_square=self.values[0].l2_norm_square())
        for k in [k for k, v in host.items() if torch.is_tensor(v)]:      # clipped runs: the norm is a device scalar
            scal[k] = host.pop(k).detach()
        keys = list(scal)
        packed = torch.stack([scal[k].reshape(()).float() for k in keys])
        if self._graph is not None:                              # captured update: a D2H node into the graph's static pinned buffer
            return self._graph.log_node(keys, packed, host, dict(real_batch_size=batch_size, real_batch_traj_num=rows,
                                                                  average_traj_len=self.replay_buffer.size / len(self.replay_buffer),
                                                                  amp_scalar_pi=0, amp_scalar_q=0))
        if ops.AMAX_VERIFY and self.device.type == 'cuda':       # RESEL_AMAX_VERIFY=1: every mode-2 product of this update had its handles checked
            ops.amax_verify_raise(self.device)
        log = DeferredLog(keys, packed, pinned=self.device.type == 'cuda')     # ONE device->host copy of all scalars
        log.set_host({k: float(v) for k, v in host.items()})
        log.set_host(dict(real_batch_size=batch_size, real_batch_traj_num=rows,
                          average_traj_len=self.replay_buffer.size / len(self.replay_buffer), amp_scalar_pi=0, amp_scalar_q=0))
        if not getattr(self, 'defer_log', False):
            log.resolve()                                        # reference behaviour: floats in hand when the call returns
        return log
