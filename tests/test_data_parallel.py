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
    torch.save(dict(policy=alg.policy.store.flat.clone(), value=alg.values[0].store.flat.clone(), guard=alg.Q_guard.state.detach().clone(),
                    alpha=alg.log_sac_alpha.detach().clone(), critic_loss=log['critic_loss'], calls=dict(alg.grad_sync.calls)),
               os.path.join(out_dir, f'rank{rank}.pt'))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize('mode,guard', [('different', 'bucket'), ('same', 'bucket'), ('union', 'bucket'), ('union', 'allreduce')])
def test_two_rank_update(tmp_path, mode, guard, monkeypatch):
    """guard = bucket (default): one collective per optimizer step, the ranks' Q-guard extrema ride in the critic's gradient bucket
    (4 floats per rank in a zero-filled tail); allreduce: two MAX all-reduces inside the target.  Either way the parameters AND the
    guard state equal the single-process update over the union batch."""
    monkeypatch.setenv('RESEL_DP_GUARD', guard)
    mp.spawn(_worker, args=(2, _free_port(), mode, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, f'rank{i}.pt')) for i in range(2))
    assert r0['calls']['all_reduce_sum'] == 4 and r0['calls']['all_reduce_max'] == (0 if guard == 'bucket' else 4), r0['calls']
    for k in ('policy', 'value', 'alpha', 'guard'):
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
        np.testing.assert_allclose(r0['guard'], alg.Q_guard.state.detach(), rtol=1e-6)


# ---- the launcher: `python bench.py --gpus N` starts its own ranks; the torchrun form keeps working (no GPU needed: --spawn-dry-run) ----

ROOT = os.path.dirname(HERE)


def _bench(args, env=None, launcher=()):
    import json
    import subprocess
    cmd = [sys.executable, *launcher, os.path.join(ROOT, 'bench.py'), *args]
    e = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    e.update(env or {})
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=e)
    js = [l for l in r.stdout.splitlines() if l.startswith('{')]
    return r, (json.loads(js[-1]) if js else None)


def test_bench_starts_its_own_ranks_and_splits_a_global_batch():
    r, line = _bench(['--gpus', '2', '--global-rows', '128', '--steps', '7', '--warmup', '2', '--spawn-dry-run'])
    assert r.returncode == 0, r.stderr[-2000:]
    assert line['n_gpus'] == 2 and line['rows_per_rank'] == 64 and line['global_rows'] == 128 and line['scaling'] == 'strong'
    assert line['local_rank_sum'] == 1 and line['rccl_ranks'] == 2 and line['collectives']['all_reduce_sum'] == 1
    assert line['steps'] == 7 and line['warmup'] == 2 and 'spawned its own ranks' in line['launcher']
    assert len([l for l in r.stdout.splitlines() if l.startswith('{')]) == 1          # ONE JSON line on stdout


def test_bench_weak_scaling_rows_per_rank_under_the_launcher():
    r, line = _bench(['--gpus', '4', '--spawn-dry-run'])
    assert r.returncode == 0, r.stderr[-2000:]
    assert line['n_gpus'] == 4 and line['rows_per_rank'] == 64 and line['global_rows'] == 256 and line['scaling'] == 'weak'
    assert line['local_rank_sum'] == 0 + 1 + 2 + 3


def test_bench_exits_non_zero_when_a_rank_fails():
    import time
    t0 = time.time()
    r, line = _bench(['--gpus', '2', '--spawn-dry-run'], env={'RESEL_BENCH_DRY_FAIL_RANK': '1'})
    assert r.returncode != 0 and line is None and 'rank 1 failed' in r.stderr
    assert time.time() - t0 < 120                    # the surviving rank (blocked in the rendezvous) was taken down, not waited for
    r, line = _bench(['--gpus', '3', '--global-rows', '128', '--spawn-dry-run'])
    assert r.returncode != 0 and line is None and 'does not split' in r.stderr


def test_bench_under_torchrun_keeps_working():
    r, line = _bench(['--gpus', '2', '--global-rows', '32', '--spawn-dry-run'],
                     launcher=('-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                               '--master-port', str(_free_port())))
    assert r.returncode == 0, r.stderr[-2000:]
    assert line['n_gpus'] == 2 and line['rows_per_rank'] == 16 and 'launcher' not in line


def test_one_rank_group_issues_and_counts_collectives(monkeypatch):
    """RESEL_DP_FORCE_COLLECTIVES=1: init_from_env creates a ONE-rank group and GradSync issues (and counts) every collective in it."""
    from offpolicy_rnn.parallel.data_parallel import GradSync, init_from_env
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
        monkeypatch.delenv(k, raising=False)
    assert not GradSync().active
    monkeypatch.setenv('RESEL_DP_FORCE_COLLECTIVES', '1')
    assert init_from_env(backend='gloo') == (0, 1, 0) and dist.is_initialized()
    try:
        gs = GradSync()
        assert gs.active and gs.world == 1 and gs.backend == 'gloo'
        t = torch.arange(6.0)
        gs.all_reduce_async_(t)
        gs.wait()
        gs.all_reduce_max_(t[:2])
        gs.broadcast_(t)
        assert torch.equal(t, torch.arange(6.0))
        assert gs.calls == dict(all_reduce_sum=1, all_reduce_max=1, broadcast=1) and gs.bytes['all_reduce_sum'] == 24
        gs.reset_counters()
        assert sum(gs.calls.values()) == 0
    finally:
        dist.destroy_process_group()
