"""Nested wall-clock timer with the reference's tag interface (offpolicy_rnn/utility/timer.py:5-68)."""
import time
from collections import defaultdict


class Timer:
    def __init__(self):
        self._open = {}
        self._acc = defaultdict(float)

    def register_point(self, tag='default', level=1):
        self._open[level] = (tag, time.time())

    def register_end(self, level=1):
        tag, t0 = self._open.pop(level, (None, None))
        if tag is not None:
            self._acc[tag] += time.time() - t0

    def summary(self, summation=True):
        out = dict(self._acc)
        self._acc.clear()
        return out
