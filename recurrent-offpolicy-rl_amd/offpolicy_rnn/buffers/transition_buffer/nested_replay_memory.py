"""Packed full-trajectory batch producer (reference offpolicy_rnn/buffers/transition_buffer/nested_replay_memory.py).

`sample_trajs` returns the batch layout every kernel relies on (SURVEY.md section 8(c) "verified batch layout"):
each sampled trajectory occupies `skip_step` leading slots (the last of them is the PRE-STEP slot carrying s_0 as
next_state) followed by its transitions; several short trajectories can share one row (first-fit packing) and are
separated by `start = 1`; rows are trimmed to the longest packed length + 1.  The whole batch is ONE row-major fp32
array (rows, T', width) - fields are column views - so the trainer ships it to the GPU with a single copy.
"""
import math
from typing import Tuple

import numpy as np

from .replay_memory import MemoryArray, Transition


class NestedMemoryArray(MemoryArray):
    def __init__(self, max_transition_num: int = 1000, max_traj_step: int = 1000, rnn_slice_length: int = 1,
                 additional_history_len: int = 0, map_to_two_power=True):
        cap = max_traj_step + 2 + additional_history_len
        if map_to_two_power:
            cap = self.nearest_power_of_two(cap)
            if cap >= 2048:
                print(f'[ WARNING ] row capacity is {cap}; consider map_to_two_power=False')
        super().__init__(max_transition_num, cap, rnn_slice_length)
        self._additional_history_len = additional_history_len
        self._skip_step = 1 + additional_history_len
        self._stage = None                                    # reusable (rows, cap, width) staging array

    @staticmethod
    def nearest_power_of_two(x):
        return int(math.ceil(2 ** max(int(math.ceil(math.log(x, 2))), 0)))

    def load_equalize(self, traj_lens, max_traj_length):
        """First-fit packing into rows of `max_traj_length` slots: a trajectory goes to the open row that it leaves
        with the least spare room (strictly larger than the trajectory), else opens a new row."""
        rows, spare = [], []
        for idx, n in enumerate(traj_lens):
            best, best_left = -1, max_traj_length + 1
            for r, room in enumerate(spare):
                if room > n and room - n < best_left:
                    best, best_left = r, room - n
            if best >= 0:
                rows[best].append(idx)
                spare[best] = best_left
            else:
                rows.append([idx])
                spare.append(max_traj_length - n)
        return rows

    def get_equalized_valid_num_each_traj(self, traj_len_added_1, desired_total_valid_number):
        order = np.argsort(traj_len_added_1)
        n = len(traj_len_added_1)
        avg = int(np.ceil(desired_total_valid_number / n))
        out, got = [avg] * n, 0
        for i in range(n):
            length = traj_len_added_1[order[i]] - 1
            want = int(np.ceil((desired_total_valid_number - got) / (n - i)))
            if want <= 0:
                want = avg
            want = min(want, length)
            got += want
            out[order[i]] = want
        return out

    def _mask_rnd_select(self, mask, select_num):
        flat = mask.reshape((-1,))
        idx = flat.nonzero()[0]
        flat[idx[np.random.permutation(idx.shape[0])[:-select_num]]] = 0

    def sample_trajs(self, batch_size, max_sample_size=None, get_all=False, randomize_mask=False,
                     valid_number_post_randomized=0, equalize_data_of_each_traj=False, random_trunc_traj=False,
                     copy=False, nest_stack_trajs=True) -> Tuple[Transition, int, np.ndarray, np.ndarray]:
        skip = self._skip_step
        if get_all:
            picked = np.arange(self.available_traj_num)
        else:
            if random_trunc_traj:
                batch_size *= 2
            picked = self._traj_ind_sample(batch_size, max_sample_size)
        if random_trunc_traj:
            lens = [np.random.randint(0, self.trajectory_length[i]) + 1 + skip for i in picked]
        else:
            lens = [self.trajectory_length[i] + skip for i in picked]
        starts = [self.trajectory_start[i] for i in picked]
        valid_nums = None
        if randomize_mask and equalize_data_of_each_traj:
            valid_nums = self.get_equalized_valid_num_each_traj(lens, valid_number_post_randomized)
        groups = self.load_equalize(lens, self.max_traj_step) if nest_stack_trajs else [[i] for i in range(len(lens))]
        nrow = len(groups)
        total_size = int(sum(lens) - len(lens) * skip)

        if self._stage is None or self._stage.shape[0] < nrow:
            self._stage = np.zeros((nrow, self.max_traj_step, self.width), dtype=self.STORE_DTYPE)
        stage = self._stage
        need = max(sum(lens[j] for j in grp) for grp in groups) + 1     # only this prefix of every row is ever returned
        stage[:nrow, :need] = 0
        valid = np.zeros((nrow, need, 1), dtype=self.STORE_DTYPE)
        R = self.name2range
        tgt = np.r_[R['next_state'][0]:R['next_state'][1], R['reward'][0]:R['reward'][1], R['state'][0]:R['state'][1]]
        src = np.r_[R['state'][0]:R['state'][1], R['reward_input'][0]:R['reward_input'][1], R['last_state'][0]:R['last_state'][1]]
        a0, a1 = R['action']
        m0, s0 = R['mask'][0], R['start'][0]
        longest, table = 0, []
        for r, grp in enumerate(groups):
            pos, seq = 0, [1]                                   # the leading dummy length-1 sequence (appendix D.9)
            for j in grp:
                n = lens[j]
                body = self.memory_buffer[starts[j]:starts[j] + n - skip]
                seq.append(n)
                stage[r, pos + skip:pos + n] = body
                stage[r, pos + skip - 1, tgt] = body[0, src]   # pre-step slot: next_state<-s0, reward<-r_in0, state<-last_state0
                stage[r, pos + skip - 1, a0:a1] = 0
                stage[r, pos:pos + skip, s0] = 1
                valid[r, pos + skip:pos + n, 0] = body[:, m0]
                if valid_nums is not None:
                    zero = np.random.permutation(n - skip)[:-valid_nums[j]] + pos + skip
                    stage[r, zero, m0] = 0
                pos += n
            longest = max(longest, pos)
            stage[r, pos:need, s0] = 1                          # trailing padding is "start" everywhere
            table.append(seq)
        longest += 1
        width = max(len(s) for s in table)
        traj_len_array = np.zeros((nrow, width))
        for r, seq in enumerate(table):
            traj_len_array[r, :len(seq)] = seq
        view = stage[:nrow, :longest]
        if copy:
            view = view.copy()
        result = self.array_to_transition(view)
        valid = valid[:, :longest]
        if randomize_mask and not equalize_data_of_each_traj:
            self._mask_rnd_select(result.mask, valid_number_post_randomized)
        self._last_batch_array = view                           # the single array behind every field view
        self._last_batch_shape = view.shape[:2]
        return result, total_size, valid, traj_len_array

    # ------------------------------------------------------------------ device-resident variant
    def device_supported(self, randomize_mask=False, **_):
        """The device packer covers every sampling mode whose randomness lives in the PLAN (which trajectories, truncated
        lengths, row packing); per-transition mask randomisation stays on the host path."""
        return not randomize_mask

    def _mirror(self, device):
        """Device copy of the ring, refreshed for the rows written since the last call (a rollout adds one trajectory
        between updates; the synthetic benchmark fills the ring once)."""
        import torch
        st = self.__dict__.setdefault('_dev_state', {'buf': None})
        dirty = getattr(self, '_dirty', None)
        if st['buf'] is None or st['buf'].device != device or st['buf'].shape != self.memory_buffer.shape:
            st['buf'] = torch.from_numpy(self.memory_buffer).to(device)
        elif dirty is None or len(dirty) > 64:       # whole ring again, IN PLACE: captured updates hold the mirror's address
            st['buf'].copy_(torch.from_numpy(self.memory_buffer))
        else:
            for s0, n in dirty:                      # through pinned memory: a pageable copy would stall the launch queue
                rows = torch.from_numpy(self.memory_buffer[s0:s0 + n])
                st['buf'][s0:s0 + n].copy_(rows.pin_memory() if device.type == 'cuda' else rows, non_blocking=True)
        self._dirty = []
        return st['buf']

    def plan_trajs_device(self, batch_size, max_sample_size=None, get_all=False, random_trunc_traj=False, nest_stack_trajs=True):
        """Host half of `sample_trajs_device`: the same sampling decisions (and numpy RNG consumption) as `sample_trajs`.  Returns a dict
        with the int32 plan `seg` [nseg, 4] = (row, first slot, length incl. skip, first transition) and the scalars the gather needs."""
        skip = self._skip_step
        if get_all:
            picked = np.arange(self.available_traj_num)
        else:
            if random_trunc_traj:
                batch_size *= 2
            picked = self._traj_ind_sample(batch_size, max_sample_size)
        if random_trunc_traj:
            lens = [np.random.randint(0, self.trajectory_length[i]) + 1 + skip for i in picked]
        else:
            lens = [self.trajectory_length[i] + skip for i in picked]
        starts = [self.trajectory_start[i] for i in picked]
        groups = self.load_equalize(lens, self.max_traj_step) if nest_stack_trajs else [[i] for i in range(len(lens))]
        nrow = len(groups)
        total_size = int(sum(lens) - len(lens) * skip)
        plan, table, longest = [], [], 0
        for r, grp in enumerate(groups):
            pos, seq = 0, [1]
            for j in grp:
                plan.append((r, pos, lens[j], starts[j]))
                seq.append(lens[j])
                pos += lens[j]
            longest = max(longest, pos)
            table.append(seq)
        longest += 1
        traj_len_array = np.zeros((nrow, max(len(s) for s in table)))
        for r, seq in enumerate(table):
            traj_len_array[r, :len(seq)] = seq
        seg = np.asarray(plan, dtype=np.int32)
        return dict(seg=seg, max_len=int(seg[:, 2].max()), nrow=nrow, longest=longest, total_size=total_size, table=traj_len_array)

    def _gather_pairs(self, device):
        import torch
        st = self.__dict__.setdefault('_dev_plan', {})
        if st.get('device') != device:
            from ...utility.pinned import PinnedRing
            R = self.name2range
            pairs = []
            for dst, src in (('next_state', 'state'), ('reward', 'reward_input'), ('state', 'last_state')):
                pairs += list(zip(range(*R[dst]), range(*R[src])))
            st.update(device=device, pairs=torch.tensor(pairs, dtype=torch.int32).to(device), ring=PinnedRing(torch.int32, depth=4))
        return st

    def gather_planned(self, device, seg_dev, max_len, nrow, longest):
        """Device half: the packed batch [rows, T', W + 3] assembled on the GPU from the device mirror of the ring and the plan."""
        from ...hip import ops
        R = self.name2range
        st = self._gather_pairs(device)
        out = ops.gather_trajs(self._mirror(device), seg_dev, max_len, self._skip_step, nrow, longest, R['mask'][0], R['start'][0],
                               R['done'][0], R['timeout'][0] if R['timeout'][1] > R['timeout'][0] else -1, st['pairs'])
        self._last_batch_shape = (nrow, longest)
        return out

    def sample_trajs_device(self, device, batch_size, max_sample_size=None, get_all=False, random_trunc_traj=False,
                            nest_stack_trajs=True):
        """Same sampling decisions (and numpy RNG consumption) as `sample_trajs`, but the batch array is assembled on the
        device by `ops.gather_trajs` from the device mirror of the ring: returns (batch [rows, T', W + 3] on `device`,
        total_size, traj_len_array)."""
        import torch
        pl = self.plan_trajs_device(batch_size, max_sample_size, get_all, random_trunc_traj, nest_stack_trajs)
        seg = pl['seg']
        st = self._gather_pairs(device)
        # the plan block is rewritten on every sample while earlier copies may still be queued: one event per block
        host = st['ring'].stage(seg.size, device, 1024).view(-1, 4)
        host.copy_(torch.from_numpy(seg))
        seg_dev = st['ring'].upload(host, device)
        out = self.gather_planned(device, seg_dev, pl['max_len'], pl['nrow'], pl['longest'])
        return out, pl['total_size'], pl['table']
