"""world_size-2 data parallelism with the REAL HIP kernels: two processes share cuda:0 and exchange the flat gradient buffer
through `gloo` (it stages CUDA tensors through the host; RCCL needs one GPU per rank: the two-rank `nccl` test below runs where two
GPUs exist and skips itself on a one-GPU box, where the one-rank RCCL group test is what runs).  Two ranks holding the SAME rows and the same actor noise must reproduce the single-process update -
this pins the masked-sum losses + piggy-backed valid count + flat AdamW normalisation on the device path; two ranks holding
DISJOINT trajectory sets (each trains on all of its own, actor noise off) must reproduce the single-process update over the
UNION batch - this pins the shared REDQ subset stream and the global Q-guard (phased `resel_sac_target_phase`)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.gpu


LENS = (12, 5, 7, 12, 9)
SPLIT = ((0, 2, 4), (1, 3))


class _Patch:                                        # worker processes patch for good; the pytest process passes monkeypatch
    def setattr(self, obj, name, val):
        setattr(obj, name, val)


def _build(rnn, keep=None, batch=30, quiet=False, patcher=None):
    sys.path[:0] = [HERE, os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), 'recurrent-offpolicy-rl_amd')]
    from test_host_logic import _push, _synth, make_parameter
    from offpolicy_rnn import alg_init
    from offpolicy_rnn.utility import rng
    if quiet:
        (patcher or _Patch()).setattr(rng, 'randn', lambda shape, device, dtype=torch.float32: torch.zeros(tuple(shape), dtype=dtype, device=device))
    torch.manual_seed(0)
    np.random.seed(0)
    alg = alg_init(make_parameter(rnn, sac_batch_size=batch, cuda_inference=True))
    rs = np.random.RandomState(3)
    for i, n in enumerate(LENS):
        o, a, r = _synth(rs, n, 5, 3)
        if keep is None or i in keep:
            _push(alg.replay_buffer, o, a, r, early_done=(n != 12))
    torch.manual_seed(11)
    torch.cuda.manual_seed_all(11)
    np.random.seed(11)
    alg._subset_rng = np.random.RandomState(int(alg.parameter.seed) + 7919)     # the stream data-parallel ranks share
    return alg


def _run(alg, steps=2, graph=False):
    step = alg.train_one_batch
    if graph:                                        # every update through GraphedUpdate.step(): eager warm-up, then recorded + replayed
        from offpolicy_rnn.algorithm.graphed_update import GraphedUpdate
        assert GraphedUpdate.refusal(alg) is None, GraphedUpdate.refusal(alg)
        gu = GraphedUpdate(alg, warmup=1)
        step = gu.step
    for _ in range(steps):
        log = step()
        alg.grad_num += 1
    torch.cuda.synchronize()
    if graph:
        assert len(gu.graphs) >= 1 and gu.eager_fallbacks < steps, (len(gu.graphs), gu.eager_fallbacks)
        if alg.grad_sync.active:                     # cut at the two gradient exchanges: three graphs per update with an actor step
            assert max(len(g['segs']) for g in gu.graphs.values()) == 3
        gu.close()
    return dict(policy=alg.policy.store.flat[:alg.policy.store.numel].detach().cpu(),
                value=alg.values[0].store.flat[:alg.values[0].store.numel].detach().cpu(),
                alpha=alg.log_sac_alpha.detach().cpu(), critic_loss=log['critic_loss'], guard=alg.Q_guard.state.detach().cpu())


def _worker(rank, world, port, rnn, out_dir, union=False, backend='gloo', graph=False, steps=2, per=1):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(rank if backend == 'nccl' else 0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    alg = _build(rnn, keep=SPLIT[rank], batch=sum(LENS[i] for i in SPLIT[rank]), quiet=True) if union else _build(rnn)
    alg._subset_rng = None                             # product default under world > 1: the shared stream
    alg.parameter.policy_update_per = per
    alg.grad_sync.__init__()
    assert alg.grad_sync.world == world and alg.device.type == 'cuda' and alg.grad_sync.backend == backend
    res = _run(alg, steps, graph)
    res['calls'] = dict(alg.grad_sync.calls)
    res['guard'] = alg.Q_guard.state.detach().cpu()
    torch.save(res, os.path.join(out_dir, f'rank{rank}.pt'))
    dist.destroy_process_group()


@pytest.mark.parametrize('rnn', ['smamba_s8_c4_b1_nln', 'gilr'])
def test_two_ranks_on_one_gpu_reproduce_the_single_process_update(tmp_path, rnn):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from test_data_parallel import _free_port
    mp.spawn(_worker, args=(2, _free_port(), rnn, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, f'rank{i}.pt')) for i in range(2))
    for k in ('policy', 'value', 'alpha'):
        assert torch.equal(r0[k], r1[k]), f'{k} diverged across ranks'
    ref = _run(_build(rnn))                          # single process, same rows, same noise
    for k in ('policy', 'value', 'alpha'):
        np.testing.assert_allclose(r0[k].numpy(), ref[k].numpy(), rtol=2e-4, atol=2e-6, err_msg=k)


@pytest.mark.parametrize('guard', ['bucket', 'allreduce'])
@pytest.mark.parametrize('rnn', ['smamba_s8_c4_b1_nln', 'gilr'])
def test_two_ranks_with_disjoint_rows_reproduce_the_union_batch_update(tmp_path, rnn, guard, monkeypatch):
    """guard = bucket (default): ONE collective per optimizer step - the ranks' Q-guard extrema ride in the critic's gradient bucket;
    allreduce: the three-phase target with two MAX all-reduces.  Both give the single-process update over the union batch and its
    guard state (the guard acts on the next update's target only)."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from test_data_parallel import _free_port
    monkeypatch.setenv('RESEL_DP_GUARD', guard)
    mp.spawn(_worker, args=(2, _free_port(), rnn, str(tmp_path), True), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, f'rank{i}.pt')) for i in range(2))
    for k in ('policy', 'value', 'alpha', 'guard'):
        assert torch.equal(r0[k], r1[k]), f'{k} diverged across ranks'
    # 2 updates x (critic step + actor step) flat-gradient all-reduces; the guard: nothing of its own / 2 x 2 MAX all-reduces
    assert r0['calls']['all_reduce_sum'] == 4 and r0['calls']['all_reduce_max'] == (0 if guard == 'bucket' else 4), r0['calls']
    ref = _run(_build(rnn, batch=sum(LENS), quiet=True, patcher=monkeypatch))       # one process, all five trajectories in one batch
    for k in ('policy', 'value', 'alpha', 'guard'):
        np.testing.assert_allclose(r0[k].numpy(), ref[k].numpy(), rtol=2e-4, atol=2e-6, err_msg=k)


def test_two_ranks_over_rccl_reproduce_the_union_batch_update(tmp_path, monkeypatch):
    """Two GPUs, backend `nccl` (= RCCL): the exchange-stream branch of `all_reduce_async_` with a real peer.  Skips itself on a
    one-GPU box (RCCL refuses two ranks on one device)."""
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs (RCCL: one rank per device)')
    from test_data_parallel import _free_port
    rnn = 'smamba_s8_c4_b1_nln'
    mp.spawn(_worker, args=(2, _free_port(), rnn, str(tmp_path), True, 'nccl'), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, f'rank{i}.pt')) for i in range(2))
    for k in ('policy', 'value', 'alpha'):
        assert torch.equal(r0[k], r1[k]), f'{k} diverged across ranks'
    assert r0['calls']['all_reduce_sum'] == 4 and r0['calls']['all_reduce_max'] == 0, r0['calls']
    ref = _run(_build(rnn, batch=sum(LENS), quiet=True, patcher=monkeypatch))
    for k in ('policy', 'value', 'alpha', 'guard'):
        np.testing.assert_allclose(r0[k].numpy(), ref[k].numpy(), rtol=2e-4, atol=2e-6, err_msg=k)


@pytest.mark.parametrize('rnn,per', [('smamba_s8_c4_b1_nln', 1), ('gilr', 2)])
def test_two_ranks_graphed_update_reproduces_the_union_batch_update(tmp_path, rnn, per, monkeypatch):
    """Data parallelism WITH the update graph: each rank drives its updates through GraphedUpdate.step() - the recording is cut at the
    two gradient exchanges, so an update is three graph replays with the all-reduces (gloo here: two ranks share cuda:0) issued
    eagerly between them.  Five updates (eager warm-up, then recorded and replayed; per = 2: the graphs with and without the actor
    step alternate) of two ranks with disjoint rows against the single-process EAGER update over the union batch."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from test_data_parallel import _free_port
    steps = 5 if per == 1 else 8
    mp.spawn(_worker, args=(2, _free_port(), rnn, str(tmp_path), True, 'gloo', True, steps, per), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, f'rank{i}.pt')) for i in range(2))
    for k in ('policy', 'value', 'alpha', 'guard'):
        assert torch.equal(r0[k], r1[k]), f'{k} diverged across ranks'
    n_actor = (steps + per - 1) // per
    assert r0['calls']['all_reduce_sum'] == steps + n_actor and r0['calls']['all_reduce_max'] == 0, r0['calls']
    one = _build(rnn, batch=sum(LENS), quiet=True, patcher=monkeypatch)
    one.parameter.policy_update_per = per
    ref = _run(one, steps)
    for k in ('policy', 'value', 'alpha', 'guard'):
        np.testing.assert_allclose(r0[k].numpy(), ref[k].numpy(), rtol=5e-4, atol=5e-6, err_msg=k)


def _one_rank_bench(launcher, extra_env=None):
    import json
    import subprocess
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    env.update(RESEL_DP_FORCE_COLLECTIVES='1', MASTER_ADDR='127.0.0.1', **(extra_env or {}))
    cmd = [sys.executable, *launcher, os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1', '--rows', '8',
           '--no-cpu-baseline', '--no-strict-leg', '--no-rccl-leg', '--no-suite']
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])


def _assert_collectives_ran(line):
    assert line['n_gpus'] == 1 and np.isfinite(line['value']) and line['value'] > 0
    assert line['backend'] == 'nccl' and line['rccl_ranks'] == 1
    assert line['graph_update'] is True, line['launch']     # data-parallel groups run the update graph too (cut at the exchanges)
    # per update: critic step + actor step = 2 flat-gradient all-reduces; the target's Q-guard = 2 MAX all-reduces
    assert line['collectives_per_step']['all_reduce_sum'] == 2 and line['collectives_per_step']['all_reduce_max'] == 0, line['collectives_per_step']
    assert line['parameter_broadcasts'] >= 3 and line['collective_bytes_per_step']['all_reduce_sum'] > 1e6


def test_rccl_collectives_in_a_one_rank_group_under_torchrun():
    """The RCCL plumbing on real hardware with the one GPU a test box has: `bench.py` under torch.distributed.run with one rank and
    RESEL_DP_FORCE_COLLECTIVES=1 forms a one-rank `nccl` (= RCCL) group and ISSUES every collective of the data-parallel update -
    parameter broadcast, the flat-gradient all-reduce on the exchange stream, the Q-guard's two MAX all-reduces inside the target.
    They are identities in a one-rank group; the bench line reports how many were issued (counted at the call sites)."""
    from test_data_parallel import _free_port
    line = _one_rank_bench(('-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
                            '--master-port', str(_free_port())))
    _assert_collectives_ran(line)


def test_rccl_collectives_in_a_one_rank_group_without_a_launcher():
    """Same, started as a plain `python bench.py --gpus 1`: init_from_env creates the one-rank group itself (free local port)."""
    _assert_collectives_ran(_one_rank_bench(()))


def test_bench_line_without_a_group_reports_no_collectives():
    import json
    import subprocess
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'RESEL_DP_FORCE_COLLECTIVES')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '2', '--warmup', '1', '--rows', '8', '--no-cpu-baseline',
                        '--no-strict-leg', '--no-rccl-leg', '--no-suite'], capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert line['rccl_ranks'] == 0 and line['backend'] is None and sum(line['collectives_per_step'].values()) == 0


def test_bench_launcher_end_to_end_with_two_ranks_sharing_the_gpu():
    """`python bench.py --gpus 2` end to end on a one-GPU box: the launcher starts two ranks, both run the real kernels on cuda:0 and
    exchange gradients through `gloo` (RESEL_DP_BACKEND=gloo: RCCL refuses two ranks on one device), rank 0 reports the global
    numbers: twice the rows, the collectives of a 2-rank job counted."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'RESEL_DP_FORCE_COLLECTIVES')}
    env['RESEL_DP_BACKEND'] = 'gloo'
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--rows', '4', '--horizon', '128'],
                       capture_output=True, text=True, env=env, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    js = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(js) == 1
    line = json.loads(js[0])
    assert line['n_gpus'] == 2 and line['config']['global_rows'] == 8 and line['config']['parallelism'] == 'dp2'
    assert line['rccl_ranks'] == 2 and line['backend'] == 'gloo' and 'spawned its own ranks' in line['launcher']
    assert line['collectives_per_step']['all_reduce_sum'] == 2 and line['collectives_per_step']['all_reduce_max'] == 0
    assert np.isfinite(line['value']) and line['value'] > 0
    # strong scaling form: a fixed global batch split over the ranks
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--global-rows', '8', '--steps', '2', '--warmup', '1', '--horizon', '128'],
                       capture_output=True, text=True, env=env, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert line['scaling'] == 'strong' and line['config']['global_rows'] == 8 and 'B=4/GPU' in line['config']['workload']
