"""world_size-2 data parallelism on CPU (`gloo`): the flat-gradient all-reduce with the piggy-backed valid count.

(a) ranks that hold DIFFERENT rows end every update with identical parameters;
(b) two ranks that hold the SAME rows reproduce the single-process update (sum of two equal gradients over twice the
    count = the single-process mean gradient), which pins the global `/ valid_num` normalisation.
Kernels are the CPU oracle stand-ins (tests/oracle_backend.py); RCCL itself is exercised by the driver's multi-GPU bench."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


class _Patch:
    def setattr(self, obj, name, val):
        setattr(obj, name, val)


def _build(seed_data, same_noise_seed, patcher=None):
    sys.path[:0] = [HERE, os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), 'recurrent-offpolicy-rl_amd')]
    import oracle_backend
    oracle_backend.install(patcher or _Patch())      # worker processes patch for good; the pytest process uses monkeypatch
    from test_host_logic import _push, _synth, make_parameter
    from offpolicy_rnn import alg_init
    torch.manual_seed(0)
    np.random.seed(0)
    alg = alg_init(make_parameter('gilr', sac_batch_size=30))
    rs = np.random.RandomState(seed_data)
    for n in (12, 5, 7, 12, 9):
        o, a, r = _synth(rs, n, 5, 3)
        _push(alg.replay_buffer, o, a, r, early_done=(n != 12))
    torch.manual_seed(same_noise_seed)
    np.random.seed(same_noise_seed)
    return alg


def _worker(rank, world, port, same_data, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    alg = _build(seed_data=3 if same_data else 3 + rank, same_noise_seed=11 if same_data else 11 + rank)
    alg.grad_sync.__init__()
    assert alg.grad_sync.world == world
    for _ in range(2):
        log = alg.train_one_batch()
        alg.grad_num += 1
    torch.save(dict(policy=alg.policy.store.flat.clone(), value=alg.values[0].store.flat.clone(),
                    alpha=alg.log_sac_alpha.detach().clone(), critic_loss=log['critic_loss']), os.path.join(out_dir, f'rank{rank}.pt'))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize('same_data', [False, True])
def test_two_rank_update(tmp_path, same_data, monkeypatch):
    mp.spawn(_worker, args=(2, _free_port(), same_data, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, f'rank{i}.pt')) for i in range(2))
    for k in ('policy', 'value', 'alpha'):
        assert torch.equal(r0[k], r1[k]), f'{k} diverged across ranks'
    if same_data:
        sys.path[:0] = [HERE]
        alg = _build(seed_data=3, same_noise_seed=11, patcher=monkeypatch)   # single process, same rows, same noise
        for _ in range(2):
            alg.train_one_batch()
            alg.grad_num += 1
        n = alg.policy.store.numel
        np.testing.assert_allclose(r0['policy'][:n], alg.policy.store.flat[:n], rtol=1e-5, atol=1e-7)
        n = alg.values[0].store.numel
        np.testing.assert_allclose(r0['value'][:n], alg.values[0].store.flat[:n], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(r0['alpha'], alg.log_sac_alpha.detach(), rtol=1e-6)
