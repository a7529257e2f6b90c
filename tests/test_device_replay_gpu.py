"""Device-resident replay ring: the batch array assembled on the GPU by `resel_gather_trajs` from the
sampling plan must be BIT-EXACT the array of the host path (`sample_trajs` + `_upload_batch`, itself pinned bit-exactly to the
reference's `sample_trajs` by tests/test_host_logic.py::test_sample_trajs_equals_reference) - same numpy RNG consumption,
ragged trajectories, several trajectories packed per row, truncation, ring refresh after new pushes."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.gpu


def _trainer(lengths, rnn='gru', seed=3):
    sys.path[:0] = [HERE]
    from test_host_logic import _push, _synth, make_parameter
    from offpolicy_rnn import alg_init
    torch.manual_seed(0)
    np.random.seed(0)
    alg = alg_init(make_parameter(rnn, sac_batch_size=40, cuda_inference=True))
    rs = np.random.RandomState(seed)
    for n in lengths:
        o, a, r = _synth(rs, n, 5, 3)
        _push(alg.replay_buffer, o, a, r, early_done=(n != 12))
    return alg, rs


def _both(alg, seed, **kw):
    buf = alg.replay_buffer
    np.random.seed(seed)
    batch, size_h, valid, table_h = buf.sample_trajs(40, None, equalize_data_of_each_traj=True, **kw)
    host = alg._upload_batch(batch, valid, table_h)['state']._base        # a wide field: still a view of the batch array (the per-token scalars are a planar copy)
    np.random.seed(seed)
    dev, size_d, table_d = buf.sample_trajs_device(alg.device, 40, None, **kw)
    torch.cuda.synchronize()
    return host, dev, (size_h, table_h), (size_d, table_d)


@pytest.mark.parametrize('rnn', ['gru', 'smamba_s8_c4_b1_nln'])          # skip_step 2 and 6
@pytest.mark.parametrize('kw', [dict(nest_stack_trajs=True), dict(nest_stack_trajs=False), dict(nest_stack_trajs=True, random_trunc_traj=True)])
def test_device_batch_is_bit_exact(rnn, kw):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    alg, rs = _trainer((12, 5, 7, 12, 9, 3, 12, 6), rnn)
    for seed in (1, 2, 3):
        host, dev, h, d = _both(alg, seed, **kw)
        assert host.shape == dev.shape and h[0] == d[0] and np.array_equal(h[1], d[1])
        assert torch.equal(host, dev), f'seed {seed}: {int((host != dev).sum())} differing entries'
    # new data: the mirror must pick up rows written after it was created (ring position continues)
    from test_host_logic import _push, _synth
    for n in (4, 11):
        o, a, r = _synth(rs, n, 5, 3)
        _push(alg.replay_buffer, o, a, r, early_done=True)
    host, dev, h, d = _both(alg, 9, **kw)
    assert torch.equal(host, dev) and h[0] == d[0]


def test_update_with_device_replay_equals_host_replay():
    """Two consecutive updates with the device-built batches leave exactly the parameters of the host-built ones."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    out = []
    for device_replay in (False, True):
        alg, _ = _trainer((12, 5, 7, 12, 9), 'gilr')
        alg.device_replay = device_replay
        torch.manual_seed(11); torch.cuda.manual_seed_all(11); np.random.seed(11)
        for _ in range(2):
            alg.train_one_batch()
            alg.grad_num += 1
        torch.cuda.synchronize()
        out.append((alg.policy.store.flat.clone(), alg.values[0].store.flat.clone()))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
