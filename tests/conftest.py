import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'recurrent-offpolicy-rl_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def nested(d, prefix):
    """{'<prefix>module|key': arr} -> {module: {key: torch tensor}}"""
    import torch
    out = {}
    for k, v in d.items():
        if k.startswith(prefix):
            m, key = k[len(prefix):].split('|')
            out.setdefault(m, {})[key] = torch.from_numpy(np.array(v))
    return out


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture
def oracle_ops(monkeypatch):
    """Swap the HIP op table for the CPU oracle (host-logic tests only; see tests/oracle_backend.py)."""
    import oracle_backend
    return oracle_backend.install(monkeypatch)
