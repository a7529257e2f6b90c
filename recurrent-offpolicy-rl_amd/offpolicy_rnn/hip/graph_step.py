"""hipGraph replay of the per-environment-step policy forward (reference algorithm/sac.py:319-326).

Between two updates the outer loop calls `policy.forward` on ONE token per environment; at batch 1 that is 40-150
kernels of a few microseconds each, so the step is bound by host launch overhead, not by the GPU.  `GraphedPolicyStep`
captures the whole step once - encoders, the recurrent layer's state update, MLP head, action sampling, and the copy of
the new recurrent state over the old one - and replays it with one `hipGraphLaunch`:

  * the four inputs live in one pinned host block and one device block (a single H2D copy per step);
  * recurrent state is held in static device tensors that the graph updates in place; a cgpt KV cache takes its
    position from a device counter (`InferenceParams.device_offset`) that the graph advances;
  * the outputs (mean | sample | log-prob) land in one device block, copied back with a single D2H.

Parameters are read through their storage, so in-place optimiser steps are seen by the next replay; call
`invalidate()` after anything that re-allocates them (`load`, `.to`)."""
from typing import Optional

import numpy as np
import torch

from ..models.RNNHidden import RNNHidden


class GraphedPolicyStep:
    def __init__(self, policy, device, batch_size: int = 1, warmup: int = 2):
        if torch.device(device).type != 'cuda':
            raise RuntimeError('GraphedPolicyStep replays a hipGraph: it needs a CUDA (ROCm) device')
        self.policy, self.device, self.B, self._warmup = policy, torch.device(device), batch_size, warmup
        self._graph: Optional[torch.cuda.CUDAGraph] = None
        self._hidden: Optional[RNNHidden] = None
        self._layout = None

    # ------------------------------------------------------------------------------------------ state
    def invalidate(self):
        self._graph = None

    def _counters(self):
        return [h for h in self._hidden._data if not torch.is_tensor(h) and not isinstance(h, tuple)]

    def load_hidden(self, hidden: Optional[RNNHidden] = None):
        """Start of an episode: overwrite the static recurrent state with `hidden` (None: the zero state)."""
        if self._hidden is None:
            self._hidden = self.policy.make_init_state(self.B, self.device)
            for ip in self._counters():
                ip.device_offset = torch.zeros(1, dtype=torch.int32, device=self.device)
        for i, h in enumerate(self._hidden._data):
            src = None if hidden is None else hidden[i]
            if torch.is_tensor(h):
                h.zero_() if src is None else h.copy_(src)
            elif isinstance(h, tuple):
                for j, t in enumerate(h):
                    t.zero_() if src is None else t.copy_(src[j])
            else:                                        # KV-cache handle: stale rows beyond the position are never read
                h.reset(h.max_seqlen, h.max_batch_size)

    # ------------------------------------------------------------------------------------------ capture
    def _forward(self):
        o, a = self._layout['obs'], self._layout['act']
        x = self._in_dev.unsqueeze(1)           # [B, 1, .]: B environments, one token each (a 2-D input would be ONE sequence of length B)
        state, lst_state = x[..., :o], x[..., o:2 * o]
        lst_action, reward = x[..., 2 * o:2 * o + a], x[..., 2 * o + a:2 * o + a + 1]
        mean, _, sample, logp, new_hidden, _ = self.policy.forward(state=state, lst_state=lst_state, lst_action=lst_action,
                                                                   rnn_memory=self._hidden, reward=reward)
        self._out_dev[:, :a].copy_(mean.reshape(self.B, a))
        self._out_dev[:, a:2 * a].copy_(sample.reshape(self.B, a))
        self._out_dev[:, 2 * a:].copy_(logp.reshape(self.B, -1)[:, :1])
        for i, h in enumerate(self._hidden._data):
            if torch.is_tensor(h):
                h.copy_(new_hidden[i])
            elif isinstance(h, tuple):
                for j, t in enumerate(h):
                    t.copy_(new_hidden[i][j])

    def _capture(self, obs_dim: int, act_dim: int):
        self._layout = dict(obs=obs_dim, act=act_dim)
        width = 2 * obs_dim + act_dim + 1
        self._in_host = torch.zeros((self.B, width), dtype=torch.float32).pin_memory()
        self._in_dev = torch.zeros((self.B, width), dtype=torch.float32, device=self.device)
        self._out_dev = torch.zeros((self.B, 2 * act_dim + 1), dtype=torch.float32, device=self.device)
        self._out_host = torch.zeros((self.B, 2 * act_dim + 1), dtype=torch.float32).pin_memory()
        if self._hidden is None:
            self.load_hidden(None)
        keep = [h.clone() if torch.is_tensor(h) else None for h in self._hidden._data]
        counts = [ip.seqlen_offset for ip in self._counters()]
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side), torch.no_grad():   # eager warm-up: lazy allocations (KV caches, slopes, GEMM handles)
            for _ in range(self._warmup):
                self._forward()
        torch.cuda.current_stream(self.device).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(graph):
            self._forward()
        # the warm-up really ran (and capture advanced the host counters): put the episode state back
        for h, k in zip(self._hidden._data, keep):
            if k is not None:
                h.copy_(k)
        for ip, c in zip(self._counters(), counts):
            ip.seqlen_offset = c
            ip.device_offset.fill_(c)
        self._graph = graph

    # ------------------------------------------------------------------------------------------ step
    @torch.no_grad()
    def __call__(self, state, lst_state, lst_action, reward):
        """Numpy / CPU rows [B, dim] in -> (action_mean, action_sample, log_prob) as numpy rows.  One H2D, one graph launch,
        one D2H."""
        state = np.asarray(state, dtype=np.float32).reshape(self.B, -1)
        lst_action = np.asarray(lst_action, dtype=np.float32).reshape(self.B, -1)
        o, a = state.shape[1], lst_action.shape[1]
        if self._graph is None or self._layout != dict(obs=o, act=a):
            self._capture(o, a)
        for ip in self._counters():
            if ip.seqlen_offset >= ip.max_seqlen:
                raise RuntimeError(f'cgpt rollout: KV cache is full ({ip.seqlen_offset} tokens, max_seqlen {ip.max_seqlen})')
        buf = self._in_host.numpy()
        buf[:, :o] = state
        buf[:, o:2 * o] = np.asarray(lst_state, dtype=np.float32).reshape(self.B, -1)
        buf[:, 2 * o:2 * o + a] = lst_action
        buf[:, 2 * o + a:] = np.asarray(reward, dtype=np.float32).reshape(self.B, 1)
        self._in_dev.copy_(self._in_host, non_blocking=True)
        self._graph.replay()
        self._out_host.copy_(self._out_dev, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        for ip in self._counters():
            ip.seqlen_offset += 1
        out = self._out_host.numpy()
        return out[:, :a].copy(), out[:, a:2 * a].copy(), out[:, 2 * a:].copy()
