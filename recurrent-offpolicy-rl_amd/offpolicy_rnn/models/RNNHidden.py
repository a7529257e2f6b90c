"""RNNHidden: per-layer recurrent state plus the per-batch side channel the sequence layers read
(`rnn_start`, `mask`, `attention_concat_mask`, `grad_detach`) - reference offpolicy_rnn/models/RNNHidden.py:12-135.

Only the operations the full-trajectory trainers and the rollout loop use are provided; the slice-trainer helpers of
the reference (reshape_full_rnn_output_to_hidden, hidden_state_sample, ...) belong to out-of-scope trainers."""
import copy
from typing import List, Tuple, Union

import torch

State = Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]


def _is_rnn_type(t: str) -> bool:
    base = ('gru', 'lru', 'gilr', 'cgru', 'gilr_lstm', 'mamba', 'conv1d', 'smamba', 'transformer')
    return (t in base or (t.startswith('e') and t[1:].split('-')[0] in base) or t.startswith(('conv1d', 'econv1d', 'mamba', 'smamba', 'transformer')))


class RNNHidden:
    def __init__(self, rnn_num: int, rnn_types: List[str], device: torch.device = torch.device('cpu'), batch_first=False):
        assert len(rnn_types) == rnn_num, 'number of rnn layers should be equal to the rnn types'
        self._rnn_types = list(rnn_types)
        self._rnn_num = rnn_num
        self._data: List[State] = []
        self._device = device
        self._batch_first = batch_first
        self._rnn_start = self._attention_concat_mask = self._mask = self._grad_detach = None

    # ---- side channel -------------------------------------------------------------------------
    def set_rnn_start(self, v):
        self._rnn_start = v

    def set_attention_concat_mask(self, v):
        self._attention_concat_mask = v

    def set_mask(self, v):
        self._mask = v

    def set_grad_detach(self, v):
        self._grad_detach = v

    rnn_start = property(lambda self: self._rnn_start)
    attention_concat_mask = property(lambda self: self._attention_concat_mask)
    mask = property(lambda self: self._mask)
    grad_detach = property(lambda self: self._grad_detach)
    size = property(lambda self: len(self._data))
    device = property(lambda self: self._device)
    capacity = property(lambda self: self._rnn_num)

    def _carry_flags(self, other: 'RNNHidden') -> 'RNNHidden':
        other._rnn_start, other._attention_concat_mask = self._rnn_start, self._attention_concat_mask
        other._mask, other._grad_detach = self._mask, self._grad_detach
        return other

    # ---- container ----------------------------------------------------------------------------
    def append(self, hidden_state: State, rnn_type=None) -> None:
        assert len(self._data) < self.capacity, 'hidden num exceeds the number of RNN layers'
        if rnn_type is not None:
            assert rnn_type == self._rnn_types[self.size]
        self._data.append(hidden_state)

    def __getitem__(self, key):
        if isinstance(key, slice):
            out = RNNHidden(len(self._data[key]), self._rnn_types[key], self._device, self._batch_first)
            out._data = self._data[key]
            return self._carry_flags(out)
        return self._data[key]

    def __setitem__(self, key, value):
        self._data[key] = value

    def __len__(self) -> int:
        return len(self._data)

    def __add__(self, other):
        if other is None:
            return self
        if not isinstance(other, RNNHidden):
            return NotImplemented
        out = RNNHidden(self._rnn_num + other._rnn_num, self._rnn_types + other._rnn_types, self._device, self._batch_first)
        out._data = self._data + other._data
        return out

    @torch.no_grad()
    def init_hidden_by_type(self, rnn_type: str, batch_size: int, unit_num: int, device) -> State:
        if rnn_type == 'lstm':
            return (torch.zeros((1, batch_size, unit_num), device=device), torch.zeros((1, batch_size, unit_num), device=device))
        if _is_rnn_type(rnn_type):
            return torch.zeros((1, batch_size, unit_num), device=device)
        raise NotImplementedError(f'rnn type: {rnn_type} has not been implemented!!')

    @torch.no_grad()
    def init_random_hidden_by_type(self, rnn_type: str, batch_size: int, unit_num: int, device) -> State:
        if rnn_type == 'lstm':
            return (torch.rand((1, batch_size, unit_num), device=device) * 2 - 1, torch.rand((1, batch_size, unit_num), device=device) * 2 - 1)
        if _is_rnn_type(rnn_type):
            return torch.rand((1, batch_size, unit_num), device=device) * 2 - 1
        raise NotImplementedError(f'rnn type: {rnn_type} has not been implemented!!')

    def to_device(self, device) -> None:
        if self._device != device:
            self._device = device
            self._data = [tuple(t.to(device) for t in d) if isinstance(d, tuple) else (d.to(device) if torch.is_tensor(d) else d) for d in self._data]

    def hidden_detach_(self) -> None:
        self._data = [tuple(t.detach() for t in d) if isinstance(d, tuple) else (d.detach() if torch.is_tensor(d) else d) for d in self._data]

    def hidden_detach(self) -> 'RNNHidden':
        out = copy.deepcopy(self)
        out.hidden_detach_()
        return out

    def __copy__(self):
        out = RNNHidden(self._rnn_num, self._rnn_types, self._device, self._batch_first)
        out._data = self._data
        return self._carry_flags(out)

    def __deepcopy__(self, memo):
        out = RNNHidden(self._rnn_num, self._rnn_types, self._device, self._batch_first)
        out._data = [tuple(t.clone() for t in d) if isinstance(d, tuple) else (d.clone() if torch.is_tensor(d) else copy.deepcopy(d)) for d in self._data]
        for k in ('_rnn_start', '_attention_concat_mask', '_mask', '_grad_detach'):
            v = getattr(self, k)
            setattr(out, k, v.clone() if torch.is_tensor(v) else v)
        return out

    def __str__(self):
        return '\n'.join(f'RNN hidden {i + 1}/{len(self._data)} {getattr(d, "shape", "")}' for i, d in enumerate(self._data))
