"""QValueGuard with DEVICE-resident state (reference offpolicy_rnn/utility/q_value_guard.py:4-45).

The reference keeps min / max as Python floats and pays two `.item()` syncs in clamp() (first call) and two in
update() (every call).  Here the state is a 4-float device tensor {min, max, initialised, decay} that the fused
target kernel (`ops.sac_target`) reads and updates in place; get_min()/get_max() sync only when somebody asks."""
import torch


class QValueGuard:
    def __init__(self, guard_min=True, guard_max=True, decay_ratio=1.0, device=None):
        assert guard_min and guard_max, 'the full-trajectory trainers guard both sides'
        self._decay_ratio = decay_ratio
        self.state = torch.tensor([1000000.0, -1000000.0, 0.0, decay_ratio], dtype=torch.float32, device=device)

    def to(self, device):
        self.state = self.state.to(device)
        return self

    def reset(self):
        self.state[0], self.state[1], self.state[2] = 1000000.0, -1000000.0, 0.0

    def get_min(self) -> float:
        return float(self.state[0].item())

    def get_max(self) -> float:
        return float(self.state[1].item())
