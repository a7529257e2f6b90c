"""numpy restatement of the replay buffer's batch producer (TEST INFRASTRUCTURE ONLY).

Follows offpolicy_rnn/buffers/transition_buffer/replay_memory.py (MemoryArray) and
offpolicy_rnn/buffers/transition_buffer/nested_replay_memory.py (NestedMemoryArray.sample_trajs).
Storage here is a python list of per-trajectory float64 arrays (the reference keeps one flat ring);
the sampled batch - layout, flags, RNG consumption order - is what must be identical.
"""
import math
from collections import namedtuple
from typing import List

import numpy as np

FIELDS = ('state', 'last_state', 'last_action', 'action', 'next_state', 'reward', 'logp', 'mask', 'start',
          'done', 'reward_input', 'timeout')                       # replay_memory.py:11
Transition = namedtuple('Transition', FIELDS)


def _width(item):
    if item is None:
        return 0
    if isinstance(item, np.ndarray):
        return item.shape[-1]
    if isinstance(item, list):
        return len(item)
    return 1


class OracleBuffer:
    def __init__(self, max_transition_num=1000, max_traj_step=1000, additional_history_len=0, map_to_two_power=True):
        cap = max_traj_step + 2 + additional_history_len          # nested_replay_memory.py:11
        if map_to_two_power:
            cap = int(math.ceil(2 ** max(int(math.ceil(math.log(cap, 2))), 0)))          # :28-36
        self.max_traj_step = cap
        self.max_transition_num = max_transition_num
        self.skip = 1 + additional_history_len                    # :23
        self.trajs: List[np.ndarray] = []
        self.pending = []
        self.ranges = None

    # -- push side (replay_memory.py:183-241) ------------------------------------------------------
    def _init_ranges(self, tr):
        self.ranges, s = {}, 0
        for name, item in zip(FIELDS, tr):
            w = _width(item)
            self.ranges[name] = (s, s + w)
            s += w
        self.width = s

    def mem_push(self, tr: Transition):
        self.pending.append(tr)
        if np.all(tr.done):
            if np.all(tr.mask):
                if self.ranges is None:
                    self._init_ranges(self.pending[0])
                rows = []
                for t in self.pending:
                    parts = [np.asarray(x, dtype=np.float64).reshape(1, -1) for x in t if x is not None]
                    rows.append(np.hstack(parts))
                arr = np.vstack(rows)
                while self.size + len(arr) > self.max_transition_num:      # replay_memory.py:186-198
                    self.trajs.pop(0)
                self.trajs.append(arr)
            self.pending = []

    @property
    def size(self):
        return sum(len(t) for t in self.trajs)

    def __len__(self):
        return len(self.trajs)

    # -- sampling (replay_memory.py:56-90) ------------------------------------------------------------
    def _traj_ind_sample(self, batch_size):
        n = len(self.trajs)
        lens = [len(t) for t in self.trajs]
        desired = int(np.ceil(batch_size / (self.size / n)))
        perm = np.random.permutation(n)
        if desired <= n:
            inds = perm[:desired]
        else:
            inds = np.random.randint(0, n, (desired,))
        total = sum(lens[i] for i in inds)
        extra = []
        while total < batch_size:
            target = desired + len(extra)
            if n > target:
                idx = perm[target]
            else:
                idx = np.random.randint(low=0, high=n)
            total += lens[idx]
            extra.append(idx)
        if extra:
            inds = np.concatenate((inds, np.array(extra)), axis=0)
        return inds

    def _load_equalize(self, traj_lens):
        """first-fit-by-min-remaining bin packing (nested_replay_memory.py:38-56)."""
        cap = self.max_traj_step
        bins, room = [], []
        for idx, tl in enumerate(traj_lens):
            if bins:
                res = [r - tl if r > tl else cap + 1 for r in room]
                best = int(np.argmin(res))
                if res[best] <= cap:
                    bins[best].append(idx)
                    room[best] = res[best]
                    continue
            bins.append([idx])
            room.append(cap - tl)
        return bins

    def sample_trajs(self, batch_size, randomize_mask=False, valid_number_post_randomized=0,
                     equalize_data_of_each_traj=True, nest_stack_trajs=True):
        """nested_replay_memory.py:103-185 (random_trunc_traj=False path)."""
        R = self.ranges
        inds = self._traj_ind_sample(batch_size)
        skip = self.skip
        tlen = [len(self.trajs[i]) + skip for i in inds]
        if randomize_mask and equalize_data_of_each_traj:
            valid_nums = self._equalized_valid(tlen, valid_number_post_randomized)
        groups = self._load_equalize(tlen) if nest_stack_trajs else [[i] for i in range(len(tlen))]
        total = int(sum(tlen) - len(tlen) * skip)
        rows = np.zeros((len(groups), self.max_traj_step, self.width))
        valid = np.zeros((len(groups), self.max_traj_step, 1))
        tgt = list(range(*R['next_state'])) + list(range(*R['reward'])) + list(range(*R['state']))      # :72
        src = list(range(*R['state'])) + list(range(*R['reward_input'])) + list(range(*R['last_state']))  # :71
        m0, s0 = R['mask'][0], R['start'][0]
        a0, a1 = R['action']
        summary, real_max = [], 0
        for i, grp in enumerate(groups):
            ptr, lens = 0, [1]
            for j in grp:
                data = self.trajs[inds[j]]
                tl = tlen[j]
                lens.append(tl)
                rows[i, ptr + skip:ptr + tl] = data                                   # :160
                rows[i, ptr + skip - 1, tgt] = data[0, src]                            # :161 pre-step slot
                rows[i, ptr + skip - 1, a0:a1] = 0                                     # :162
                rows[i, ptr:ptr + skip, s0] = 1                                        # :164
                valid[i, ptr + skip:ptr + tl, 0] = data[:, m0]                         # :165
                if randomize_mask and equalize_data_of_each_traj:                      # :166-168
                    z = np.random.permutation(tl - skip)[:-valid_nums[j]] + ptr + skip
                    rows[i, z, m0] = 0
                ptr += tl
            real_max = max(real_max, ptr)
            rows[i, ptr:, s0] = 1                                                      # :171
            summary.append(lens)
        real_max += 1                                                                  # :173
        table = np.zeros((len(summary), max(len(s) for s in summary)))
        for i, s in enumerate(summary):
            table[i, :len(s)] = s
        rows = rows[:, :real_max]
        fields = {}
        for name in FIELDS:
            a, b = R[name]
            fields[name] = rows[..., a:b] if b > a else None
        return Transition(**fields), total, valid[:, :real_max], table

    @staticmethod
    def _equalized_valid(traj_len_added, desired_total):
        """nested_replay_memory.py:84-100."""
        order = np.argsort(traj_len_added)
        n = len(traj_len_added)
        avg = int(np.ceil(desired_total / n))
        out, got = [avg] * n, 0
        for i in range(n):
            tl = traj_len_added[order[i]] - 1
            want = int(np.ceil((desired_total - got) / (n - i)))
            if want <= 0:
                want = avg
            want = min(want, tl)
            got += want
            out[order[i]] = want
        return out
