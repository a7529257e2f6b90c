"""Flat transition ring with trajectory bookkeeping (reference offpolicy_rnn/buffers/transition_buffer/replay_memory.py).

Same public surface (`Transition`, `MemoryArray.mem_push / sample_transitions / size / __len__ / save_to_disk`) and the
same column layout (fields concatenated in `tuplenames` order, `None` fields have width 0).  Storage is fp32 (the
trainer converts every sampled batch to fp32 anyway, reference utility/sample_utility.py:30-31), which halves the
host traffic of the batch producer."""
import pickle
import time
from collections import namedtuple
from typing import Dict, List, Optional, Tuple

import numpy as np

tuplenames = ('state', 'last_state', 'last_action', 'action', 'next_state', 'reward', 'logp', 'mask', 'start', 'done',
              'reward_input', 'timeout')
Transition = namedtuple('Transition', tuplenames)


def _field_width(item) -> int:
    if item is None:
        return 0
    if isinstance(item, np.ndarray):
        return item.shape[-1]
    if isinstance(item, (list, tuple)):
        return len(item)
    if np.isscalar(item):
        return 1
    raise NotImplementedError(f'not implement for type of {type(item)}')


class MemoryArray:
    STORE_DTYPE = np.float32

    def __init__(self, max_transition_num: int = 1000000, max_traj_step: Optional[int] = 1000, rnn_slice_length=1):
        self.max_transition_num = int(max_transition_num)
        self.max_traj_step = max_traj_step
        self.rnn_slice_length = rnn_slice_length
        self.memory: List[Transition] = []           # transitions of the trajectory being collected
        self.memory_buffer: Optional[np.ndarray] = None
        self.name2range: Dict[str, Tuple[int, int]] = {}
        self.reset()

    def reset(self):
        self.memory = []
        self.trajectory_length: List[int] = []
        self.trajectory_start: List[int] = []
        self.ptr = 0
        self.transition_count = 0
        self._last_saving_time = 0
        self._last_saving_size = 0
        self._dirty: List[Tuple[int, int]] = []      # (first row, count) written since a device mirror last synchronised

    # ------------------------------------------------------------------ layout
    def _init_memory_buffer(self, transition: Transition):
        off = 0
        for name, item in zip(tuplenames, transition):
            w = _field_width(item)
            self.name2range[name] = (off, off + w)
            off += w
        self.width = off
        self.memory_buffer = np.zeros((self.max_transition_num + int(self.max_traj_step), off), dtype=self.STORE_DTYPE)

    @property
    def ind_range(self):
        return [list(range(*self.name2range[n])) for n in tuplenames]

    def transition_to_array(self, transition: Transition) -> np.ndarray:
        row = np.empty((1, self.width), dtype=self.STORE_DTYPE)
        for name, item in zip(tuplenames, transition):
            a, b = self.name2range[name]
            if b > a:
                row[0, a:b] = np.asarray(item, dtype=np.float64).reshape(-1)
        return row

    def array_to_transition(self, data: np.ndarray) -> Transition:
        parts = []
        for name in tuplenames:
            a, b = self.name2range[name]
            parts.append(data[..., a:b] if b > a else None)
        return Transition(*parts)

    # ------------------------------------------------------------------ push side
    def mem_push(self, transition: Transition, parallel_num=1, valid_data=True):
        if not valid_data:
            self.memory = []
            return
        self.memory.append(transition)
        if np.all(transition.done):
            if np.all(transition.mask):
                if parallel_num == 1:
                    self.complete_traj(self.memory)
                else:
                    for i in range(parallel_num):
                        self.complete_traj([Transition(*[f[i] if (f is not None and not np.isscalar(f)) else f for f in tr])
                                            for tr in self.memory])
            self.memory = []

    def complete_traj(self, memory: List[Transition]):
        if self.memory_buffer is None:
            self._init_memory_buffer(memory[0])
        n = len(memory)
        drop = 0
        count = self.transition_count
        while count + n > self.max_transition_num:            # evict the oldest trajectories
            count -= self.trajectory_length[drop]
            drop += 1
        if drop:
            self.transition_count = count
            del self.trajectory_start[:drop]
            del self.trajectory_length[:drop]
        self.trajectory_start.append(self.ptr)
        self._dirty.append((self.ptr, n))
        for tr in memory:
            self.memory_buffer[self.ptr] = self.transition_to_array(tr)[0]
            self.ptr += 1
        self.trajectory_length.append(n)
        self.transition_count += n
        if self.ptr >= self.max_transition_num:
            self.ptr = 0

    def push_trajectory(self, fields: dict):
        """Bulk insert of one finished trajectory given per-field arrays of shape [n, width] (same result as n
        `mem_push` calls; used to load synthetic / pre-recorded data without a Python loop per transition)."""
        n = len(next(iter(fields.values())))
        if self.memory_buffer is None:
            self._init_memory_buffer(Transition(*[None if fields.get(k) is None else np.asarray(fields[k]).reshape(n, -1)[0] for k in tuplenames]))
        drop, count = 0, self.transition_count
        while count + n > self.max_transition_num:
            count -= self.trajectory_length[drop]
            drop += 1
        if drop:
            self.transition_count = count
            del self.trajectory_start[:drop]
            del self.trajectory_length[:drop]
        rows = self.memory_buffer[self.ptr:self.ptr + n]
        for name in tuplenames:
            a, b = self.name2range[name]
            if b > a:
                rows[:, a:b] = np.asarray(fields[name], dtype=np.float64).reshape(n, -1)
        self.trajectory_start.append(self.ptr)
        self._dirty.append((self.ptr, n))
        self.trajectory_length.append(n)
        self.transition_count += n
        self.ptr += n
        if self.ptr >= self.max_transition_num:
            self.ptr = 0

    # ------------------------------------------------------------------ sampling
    @property
    def available_traj_num(self):
        return len(self.trajectory_length)

    def __len__(self) -> int:
        return len(self.trajectory_length)

    @property
    def size(self) -> int:
        return self.transition_count

    def _traj_ind_sample(self, batch_size, max_sample_size) -> np.ndarray:
        """Trajectory indices whose lengths sum to >= batch_size.  The numpy RNG call order is the reference's
        (replay_memory.py:56-90) so that a shared seed reproduces its batches."""
        n = self.available_traj_num
        mean_len = self.transition_count / n
        want = n if batch_size is None else int(np.ceil(batch_size / mean_len))
        cap = None
        if max_sample_size is not None:
            cap = int(np.ceil(max_sample_size / self.max_traj_step))
            want = min(want, cap)
        perm = np.random.permutation(n)
        if batch_size is None:
            picked = np.arange(n)
        elif want <= n:
            picked = perm[:want]
        else:
            picked = np.random.randint(0, n, (want,))
        total = sum(self.trajectory_length[i] for i in picked)
        extra, count = [], len(picked)
        while total < batch_size and (cap is None or count < cap):
            count += 1
            pos = want + len(extra)
            idx = perm[pos] if n > pos else np.random.randint(low=0, high=n)
            total += self.trajectory_length[idx]
            extra.append(idx)
        if extra:
            picked = np.concatenate((picked, np.array(extra)), axis=0)
        return picked

    def sample_transitions(self, batch_size: Optional[int] = None) -> Transition:
        starts = np.repeat(self.trajectory_start, self.trajectory_length)
        within = np.concatenate([np.arange(k) for k in self.trajectory_length])
        rows = starts + within
        if batch_size is not None:
            rows = rows[np.random.randint(0, self.transition_count, (batch_size,))]
        return self.array_to_transition(self.memory_buffer[rows].copy())

    def save_to_disk(self, path):
        self._last_saving_time = time.time()
        self._last_saving_size = self.size
        with open(path, 'wb') as f:
            pickle.dump(self, f, protocol=4)

    @staticmethod
    def load_from_disk(path) -> 'MemoryArray':
        with open(path, 'rb') as f:
            return pickle.load(f)
