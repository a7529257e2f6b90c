"""Reusable pinned staging buffers for asynchronous host -> device copies.

A pinned block that is rewritten on every update and shipped with `.to(device, non_blocking=True)` is only safe if the host
does not touch it again before the queued copy has run - and the trainer's launch queue runs more than one update ahead of
the device.  `PinnedRing` keeps `depth` blocks and one CUDA event per block: the event is recorded right behind the copy and
synchronised before the block is handed out again, so the host only ever waits when it is `depth` copies ahead."""
import torch


class PinnedRing:
    def __init__(self, dtype, depth: int = 3):
        self.dtype, self.depth = dtype, depth
        self._slots = []                        # [tensor, event or None]
        self._next = 0

    def stage(self, numel: int, device, min_capacity: int = 0) -> torch.Tensor:
        """A host block of `numel` elements that no queued copy reads any more (pinned when `device` is CUDA)."""
        cuda = device.type == 'cuda'
        if len(self._slots) < self.depth:
            self._slots.append([None, None])
        i = self._next % len(self._slots)
        self._next += 1
        slot = self._slots[i]
        if slot[1] is not None:
            slot[1].synchronize()
            slot[1] = None
        if slot[0] is None or slot[0].numel() < numel:
            buf = torch.empty(max(numel, min_capacity), dtype=self.dtype)
            slot[0] = buf.pin_memory() if cuda else buf
        self._current = slot
        return slot[0][:numel]

    def upload(self, host_view: torch.Tensor, device) -> torch.Tensor:
        """Asynchronous copy of a view of the block handed out by the last `stage()`; marks the block busy until it ran."""
        dev = host_view.to(device, non_blocking=True)
        if device.type == 'cuda':
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(device))
            self._current[1] = ev
        return dev
