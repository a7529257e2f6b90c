"""world_size-2 data parallelism on CPU (`gloo`): the flat-gradient all-reduce with the piggy-backed valid count, the shared
REDQ subset stream and the global Q-guard.

(a) ranks that hold DIFFERENT rows end every update with identical parameters;
(b) two ranks that hold the SAME rows reproduce the single-process update (sum of two equal gradients over twice the
    count = the single-process mean gradient), which pins the global `/ valid_num` normalisation;
(c) two ranks that hold DISJOINT trajectory sets and each train on all of theirs reproduce the single-process update over the
    UNION batch (actor noise switched off so that a trajectory sees the same draws wherever it is trained): this pins the
    semantics SURVEY.md 8(e) asks for - same critics in the REDQ minimum on every rank, Q-guard extrema of the global batch,
    global normalisation.
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


LENS = (12, 5, 7, 12, 9)


def _build(seed_data, same_noise_seed, patcher=None, keep=None, batch=30, quiet=False):
    """keep: indices of LENS this process holds (None = all); quiet: actor noise off."""
    sys.path[:0] = [HERE, os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), 'recurrent-offpolicy-rl_amd')]
    import oracle_backend
    patcher = patcher or _Patch()
    oracle_backend.install(patcher)                  # worker processes patch for good; the pytest process uses monkeypatch
    from test_host_logic import _push, _synth, make_parameter
    from offpolicy_rnn import alg_init
    from offpolicy_rnn.utility import rng
    if quiet:
        patcher.setattr(rng, 'randn', lambda shape, device, dtype=torch.float32: torch.zeros(tuple(shape), dtype=dtype, device=device))
    torch.manual_seed(0)
    np.random.seed(0)
    alg = alg_init(make_parameter('gilr', sac_batch_size=batch))
    rs = np.random.RandomState(seed_data)
    for i, n in enumerate(LENS):
        o, a, r = _synth(rs, n, 5, 3)
        if keep is None or i in keep:
            _push(alg.replay_buffer, o, a, r, early_done=(n != 12))
    torch.manual_seed(same_noise_seed)
    np.random.seed(same_noise_seed)
    alg._subset_rng = np.random.RandomState(int(alg.parameter.seed) + 7919)     # the stream the data-parallel ranks share
    return alg


SPLIT = ((0, 2, 4), (1, 3))                        # disjoint trajectory sets of the two ranks in mode 'union'


def _worker(rank, world, port, mode, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    if mode == 'union':                            # every rank trains on ALL of its own trajectories (batch = its transition count)
        alg = _build(3, 11, keep=SPLIT[rank], batch=sum(LENS[i] for i in SPLIT[rank]), quiet=True)
        alg._subset_rng = None                     # product default under world > 1: the shared stream
    else:
        same_data = mode == 'same'
        alg = _build(seed_data=3 if same_data else 3 + rank, same_noise_seed=11 if same_data else 11 + rank)
        alg._subset_rng = None
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


@pytest.mark.parametrize('mode', ['different', 'same', 'union'])
def test_two_rank_update(tmp_path, mode, monkeypatch):
    mp.spawn(_worker, args=(2, _free_port(), mode, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, f'rank{i}.pt')) for i in range(2))
    for k in ('policy', 'value', 'alpha'):
        assert torch.equal(r0[k], r1[k]), f'{k} diverged across ranks'
    if mode != 'different':
        sys.path[:0] = [HERE]
        if mode == 'same':                         # single process, same rows, same noise
            alg = _build(seed_data=3, same_noise_seed=11, patcher=monkeypatch)
        else:                                      # single process over the union of the two ranks' trajectories
            alg = _build(3, 11, patcher=monkeypatch, batch=sum(LENS), quiet=True)
        for _ in range(2):
            alg.train_one_batch()
            alg.grad_num += 1
        n = alg.policy.store.numel
        np.testing.assert_allclose(r0['policy'][:n], alg.policy.store.flat[:n], rtol=1e-5, atol=1e-7)
        n = alg.values[0].store.numel
        np.testing.assert_allclose(r0['value'][:n], alg.values[0].store.flat[:n], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(r0['alpha'], alg.log_sac_alpha.detach(), rtol=1e-6)
