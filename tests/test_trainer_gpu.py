"""End-to-end parity on the GPU box: the product's `train_one_batch` (real HIP kernels, cuda:0) against the CPU oracle
trainer on identical weights, data and noise draws.  Tolerance: north_star's fp32 bar (1e-4 rtol) is for one layer
forward; three chained optimizer steps amplify rounding, so scalars are held to 2e-3 and parameters to 1e-3 / 2e-5."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden, nested
from test_host_logic import _push, _synth, make_parameter

pytestmark = pytest.mark.gpu
from offpolicy_rnn.utility import rng as _rng_mod           # noqa: E402  (conftest put the package on sys.path)
_DEVICE_RANDN = _rng_mod.randn                               # the product's device-side draw, before the fixture below swaps it
META = json.load(open(os.path.join(GOLDEN, 'train_meta.json')))


@pytest.fixture(autouse=True)
def _cpu_noise(monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from offpolicy_rnn.utility import rng
    monkeypatch.setattr(rng, 'randn', lambda shape, device, dtype=torch.float32: torch.randn(tuple(shape), dtype=dtype).to(device))


@pytest.mark.parametrize('name', list(META))
def test_train_one_batch_gpu_vs_oracle(name):
    from offpolicy_rnn import alg_init
    from oracle.trainer import OracleTrainer, default_parameter
    from test_oracle_golden import _push as opush
    m = META[name]
    g = load_golden(f'train_{name}.npz')
    alg = alg_init(make_parameter(m['rnn'], algo=m['algo'], sac_batch_size=m['sac_batch_size']))
    assert alg.device.type == 'cuda'
    alg.policy.load_state_dict(nested(g, 'policy0|'))
    alg.values[0].load_state_dict(nested(g, 'value0|'))
    alg._value_update(tau=0.0)
    par = default_parameter(rnn=m['rnn'], D=32, algo=m['algo'], sac_batch_size=m['sac_batch_size'], policy_embedding_dim=16,
                            value_embedding_dim=16, policy_uni_model_input_mapping_dim=16, value_uni_model_input_mapping_dim=16,
                            max_buffer_transition_num=5000)
    tr = OracleTrainer(par, 5, 3, 12, smamba_semantics='gpu', policy_state=nested(g, 'policy0|'), value_state=nested(g, 'value0|'))
    rs = np.random.RandomState(9)
    for n in m['lens']:
        o, a, r = _synth(rs, n, 5, 3)
        _push(alg.replay_buffer, o, a, r, early_done=(n != 12))
        opush(tr.buffer, o, a, r, early_done=(n != 12))
    logs = []
    for runner in (alg, tr):
        torch.manual_seed(200)
        np.random.seed(200)
        out = []
        for _ in range(3):
            out.append(runner.train_one_batch())
            runner.grad_num += 1
        logs.append(out)
    for it, (a, b) in enumerate(zip(*logs)):
        for k, v in b.items():
            got = a[k][0] if isinstance(a[k], tuple) else a[k]
            want = v[0] if isinstance(v, tuple) else v
            assert got == pytest.approx(want, rel=2e-3, abs=5e-4), (it, k, got, want)
    for net, ref in ((alg.policy, tr.policy), (alg.values[0], tr.value), (alg.target_values[0], tr.target_value)):
        sd = net.state_dict()
        for mod, d in ref.items():
            for k, v in d.items():
                np.testing.assert_allclose(sd[mod][k].detach().cpu(), v.detach(), rtol=1e-3, atol=2e-5, err_msg=f'{mod}.{k}')


@pytest.mark.parametrize('rnn', ['gilr_lstm', 'conv1d_4', 'mamba_s8_c3'])
def test_train_one_batch_gpu_vs_oracle_more_layer_ids(rnn):
    """Same check for the layer ids without a trained-run fixture (their layers are pinned one by one in layers.npz /
    rollout.npz): both trainers start from the product's initial weights; two consecutive updates."""
    from offpolicy_rnn import alg_init
    from oracle.trainer import OracleTrainer, default_parameter
    from test_oracle_golden import _push as opush
    torch.manual_seed(11)
    alg = alg_init(make_parameter(rnn, sac_batch_size=40))
    cpu = lambda sd: {m: {k: v.detach().cpu().clone() for k, v in d.items()} for m, d in sd.items()}
    par = default_parameter(rnn=rnn, D=32, sac_batch_size=40, policy_embedding_dim=16, value_embedding_dim=16,
                            policy_uni_model_input_mapping_dim=16, value_uni_model_input_mapping_dim=16, max_buffer_transition_num=5000)
    tr = OracleTrainer(par, 5, 3, 12, policy_state=cpu(alg.policy.state_dict()), value_state=cpu(alg.values[0].state_dict()))
    rs = np.random.RandomState(9)
    for n in [12, 5, 7, 12, 4, 9, 6]:
        o, a, r = _synth(rs, n, 5, 3)
        _push(alg.replay_buffer, o, a, r, early_done=(n != 12))
        opush(tr.buffer, o, a, r, early_done=(n != 12))
    logs = []
    for runner in (alg, tr):
        torch.manual_seed(200)
        np.random.seed(200)
        out = []
        for _ in range(2):
            out.append(runner.train_one_batch())
            runner.grad_num += 1
        logs.append(out)
    for it, (a, b) in enumerate(zip(*logs)):
        for k, v in b.items():
            got = a[k][0] if isinstance(a[k], tuple) else a[k]
            want = v[0] if isinstance(v, tuple) else v
            assert got == pytest.approx(want, rel=2e-3, abs=5e-4), (it, k, got, want)
    for net, ref in ((alg.policy, tr.policy), (alg.values[0], tr.value)):
        sd = net.state_dict()
        for mod, d in ref.items():
            for k, v in d.items():
                np.testing.assert_allclose(sd[mod][k].detach().cpu(), v.detach(), rtol=1e-3, atol=2e-5, err_msg=f'{mod}.{k}')


@pytest.mark.parametrize('rnn', ['smamba_s8_c4_b1_nln', 'gilr'])
def test_clipped_updates_with_policy_update_per_2_vs_oracle(rnn):
    """The reference's published cadence (policy_update_per = 2, gen_tmuxp_mamba_pomdp.py:81) with gradient-norm clipping switched on
    at bounds the gradients exceed (reference :239-250, 274-287): the product folds the clip coefficient into the scale word of its flat
    AdamW kernel (no pass over the gradients, no host read); the oracle calls torch.nn.utils.clip_grad_norm_.  Four updates: logged
    scalars (the pre-clip norms among them) and parameters."""
    from offpolicy_rnn import alg_init
    from oracle.trainer import OracleTrainer, default_parameter
    from test_oracle_golden import _push as opush
    torch.manual_seed(11)
    flags = dict(policy_update_per=2, value_max_gradnorm=0.05, policy_max_gradnorm=0.01)
    alg = alg_init(make_parameter(rnn, sac_batch_size=40, **flags))
    cpu = lambda sd: {m: {k: v.detach().cpu().clone() for k, v in d.items()} for m, d in sd.items()}
    par = default_parameter(rnn=rnn, D=32, sac_batch_size=40, policy_embedding_dim=16, value_embedding_dim=16,
                            policy_uni_model_input_mapping_dim=16, value_uni_model_input_mapping_dim=16, max_buffer_transition_num=5000, **flags)
    tr = OracleTrainer(par, 5, 3, 12, smamba_semantics='gpu', policy_state=cpu(alg.policy.state_dict()), value_state=cpu(alg.values[0].state_dict()))
    rs = np.random.RandomState(9)
    for n in [12, 5, 7, 12, 4, 9, 6]:
        o, a, r = _synth(rs, n, 5, 3)
        _push(alg.replay_buffer, o, a, r, early_done=(n != 12))
        opush(tr.buffer, o, a, r, early_done=(n != 12))
    logs = []
    for runner in (alg, tr):
        torch.manual_seed(200)
        np.random.seed(200)
        out = []
        for _ in range(4):
            out.append(dict(runner.train_one_batch()))
            runner.grad_num += 1
        logs.append(out)
    for it, (a, b) in enumerate(zip(*logs)):
        assert ('actor_loss' in a) == (it % 2 == 0) and ('actor_loss' in b) == (it % 2 == 0)
        for k, v in b.items():
            got = a[k][0] if isinstance(a[k], tuple) else a[k]
            want = v[0] if isinstance(v, tuple) else v
            assert got == pytest.approx(want, rel=2e-3, abs=5e-4), (it, k, got, want)
    assert logs[0][0]['value_grad_norm'] > 0.05 and logs[0][0]['policy_grad_norm'] > 0.01      # both clips were active
    for net, ref in ((alg.policy, tr.policy), (alg.values[0], tr.value)):
        sd = net.state_dict()
        for mod, d in ref.items():
            for k, v in d.items():
                np.testing.assert_allclose(sd[mod][k].detach().cpu(), v.detach(), rtol=1e-3, atol=2e-5, err_msg=f'{mod}.{k}')


@pytest.mark.parametrize('name', ['gru_sac_discrete', 'gilr_sac_discrete'])
def test_discrete_train_one_batch_gpu_vs_reference_logs(name):
    """Discrete-action SAC-REDQ on cuda:0 against the dicts / parameters the reference itself logged (three updates)."""
    from test_host_logic import discrete_alg
    alg, g, m = discrete_alg(name)
    assert alg.device.type == 'cuda'
    torch.manual_seed(200)
    np.random.seed(200)
    for it in range(3):
        log = alg.train_one_batch()
        alg.grad_num += 1
        for k, v in m['logs'][it].items():
            got = log[k][0] if isinstance(log[k], tuple) else log[k]
            assert got == pytest.approx(v, rel=2e-3, abs=5e-4), (it, k, got, v)
    for pre, net in (('policy3|', alg.policy), ('value3|', alg.values[0]), ('target3|', alg.target_values[0])):
        sd = net.state_dict()
        for mod, d in nested(g, pre).items():
            for k, v in d.items():
                np.testing.assert_allclose(sd[mod][k].detach().cpu(), v, rtol=2e-3, atol=2e-5, err_msg=f'{pre}{mod}.{k}')


@pytest.mark.parametrize('rnn,algo,tol', [('smamba_s8_c4_b2_nln', 'sac', 1e-4), ('gilr', 'sac', 1e-4), ('cgpt_h1_l2_p0.0_ml64', 'sac', 3e-2),
                                          ('lru', 'td3', 1e-4), ('smamba_s8_c4_b1', 'td3', 1e-4), ('smamba_s8_c4_b1_nln+trunc', 'sac', 1e-4),
                                          ('gilr+trunc', 'sac', 1e-4), ('lru+rmask', 'sac', 1e-4)])
def test_shared_policy_pass_equals_two_passes_gpu(rnn, algo, tol, monkeypatch):
    """One policy forward for the target and the actor pass vs the reference's two passes, real kernels, ragged
    nested trajectories, same seeds: logged scalars and the updated policy agree (cgpt: to its bf16 attention tolerance)."""
    from offpolicy_rnn import alg_init
    runs = []
    trunc = rnn.endswith('+trunc')           # randomly truncated trajectories: segments that start in the middle of an episode
    rmask = rnn.endswith('+rmask')           # randomised loss mask: batches are built on the host path
    rnn = rnn.split('+')[0]
    for flag in ('1', '0'):
        monkeypatch.setenv('RESEL_SHARE_POLICY_PASS', flag)
        torch.manual_seed(5)
        np.random.seed(5)
        alg = alg_init(make_parameter(rnn, algo=algo, sac_batch_size=40, random_trunc_traj=trunc, randomize_mask=rmask,
                                      valid_number_post_randomized=16))
        assert alg.share_policy_pass == (flag == '1')
        rs = np.random.RandomState(9)
        for n in [12, 5, 7, 12, 4, 9, 6]:
            o, a, r = _synth(rs, n, 5, 3)
            _push(alg.replay_buffer, o, a, r, early_done=(n != 12))
        torch.manual_seed(200)
        np.random.seed(200)
        logs = []
        for _ in range(2):
            logs.append(dict(alg.train_one_batch()))
            alg.grad_num += 1
        runs.append((logs, alg.policy.store.flat.detach().clone()))
    for a, b in zip(runs[0][0], runs[1][0]):
        for k in b:
            va, vb = (a[k][0] if isinstance(a[k], tuple) else a[k]), (b[k][0] if isinstance(b[k], tuple) else b[k])
            assert va == pytest.approx(vb, rel=tol, abs=tol), k
    np.testing.assert_allclose(runs[0][1].cpu(), runs[1][1].cpu(), rtol=tol, atol=tol * 1e-1)


def test_gru_stream_overlap_equals_serial_order(monkeypatch):
    """gru: the graph-free passes run on side streams next to the critic forward / the policy forward.  Streams change the
    schedule, not the arithmetic: same seeds -> the same logs and parameters as the single-stream order, bit for bit."""
    from offpolicy_rnn import alg_init
    runs = []
    for flag in ('1', '0'):
        monkeypatch.setenv('RESEL_OVERLAP_EMBEDDING', flag)
        torch.manual_seed(5)
        np.random.seed(5)
        alg = alg_init(make_parameter('gru', sac_batch_size=60))
        assert alg.overlap_value_embedding == (flag == '1') and not alg.share_policy_pass
        rs = np.random.RandomState(9)
        for n in [12] * 8:
            o, a, r = _synth(rs, n, 5, 3)
            _push(alg.replay_buffer, o, a, r, early_done=False)
        torch.manual_seed(200)
        np.random.seed(200)
        logs = []
        for _ in range(4):
            logs.append(dict(alg.train_one_batch()))
            alg.grad_num += 1
        torch.cuda.synchronize()
        runs.append((logs, alg.policy.store.flat.detach().clone(), alg.values[0].store.flat.detach().clone()))
    for a, b in zip(runs[0][0], runs[1][0]):
        for k in b:
            va, vb = (a[k][0] if isinstance(a[k], tuple) else a[k]), (b[k][0] if isinstance(b[k], tuple) else b[k])
            assert va == vb, (k, va, vb)
    assert torch.equal(runs[0][1], runs[1][1]) and torch.equal(runs[0][2], runs[1][2])


@pytest.mark.parametrize('rnn', ['smamba_s8_c4_b1_nln', 'gru', 'cgpt_h1_l1_p0_ml32'])
def test_train_loop_end_to_end_gpu(rnn, tmp_path, monkeypatch):
    """`alg.train()` on cuda:0 with GPU sampling: graphed rollouts feed the device-mirrored replay ring, updates interleave
    every `update_interval` environment steps, the iteration-0 checkpoint is written."""
    from offpolicy_rnn import alg_init
    from test_host_logic import _short_run_parameter
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(_rng_mod, 'randn', _DEVICE_RANDN)    # graph capture needs the device generator (no host copies)
    alg = alg_init(_short_run_parameter(rnn, cuda_inference=True))
    assert alg.graph_step is not None
    alg.train()
    assert alg.graph_step._graph is not None and alg.sample_num >= 70 and alg.grad_num >= 8
    assert os.path.exists(os.path.join(alg.logger.output_dir, 'model', 'log_sac_alpha.pt'))
    assert all(torch.isfinite(p).all() for p in alg.policy.parameters()) and all(torch.isfinite(p).all() for p in alg.values[0].parameters())


@pytest.mark.parametrize('rnn', ['gilr', 'gru'])
def test_update_schedule_variants_gpu_vs_oracle(rnn):
    """utd = 2 critic updates per call with the actor updated on every second call (`policy_update_per` = 2): the shared
    policy pass / stream overlap must follow the schedule (no actor step -> plain target pass).  Four calls vs the oracle."""
    from offpolicy_rnn import alg_init
    from oracle.trainer import OracleTrainer, default_parameter
    from test_oracle_golden import _push as opush
    torch.manual_seed(11)
    over = dict(sac_batch_size=40, utd=2, policy_utd=1, policy_update_per=2)
    alg = alg_init(make_parameter(rnn, **over))
    cpu = lambda sd: {m: {k: v.detach().cpu().clone() for k, v in d.items()} for m, d in sd.items()}
    par = default_parameter(rnn=rnn, D=32, policy_embedding_dim=16, value_embedding_dim=16, policy_uni_model_input_mapping_dim=16,
                            value_uni_model_input_mapping_dim=16, max_buffer_transition_num=5000, **over)
    tr = OracleTrainer(par, 5, 3, 12, policy_state=cpu(alg.policy.state_dict()), value_state=cpu(alg.values[0].state_dict()))
    rs = np.random.RandomState(9)
    for n in [12, 5, 7, 12, 4, 9, 6, 12]:
        o, a, r = _synth(rs, n, 5, 3)
        _push(alg.replay_buffer, o, a, r, early_done=(n != 12))
        opush(tr.buffer, o, a, r, early_done=(n != 12))
    logs = []
    for runner in (alg, tr):
        torch.manual_seed(200)
        np.random.seed(200)
        out = []
        for _ in range(4):
            out.append(dict(runner.train_one_batch()))
            runner.grad_num += 1
        logs.append(out)
    for it, (a, b) in enumerate(zip(*logs)):
        assert ('actor_loss' in b) == ('actor_loss' in a), it
        for k, v in b.items():
            got = a[k][0] if isinstance(a[k], tuple) else a[k]
            want = v[0] if isinstance(v, tuple) else v
            assert got == pytest.approx(want, rel=2e-3, abs=5e-4), (it, k, got, want)
    for net, ref in ((alg.policy, tr.policy), (alg.values[0], tr.value), (alg.target_values[0], tr.target_value)):
        sd = net.state_dict()
        for mod, d in ref.items():
            for k, v in d.items():
                np.testing.assert_allclose(sd[mod][k].detach().cpu(), v.detach(), rtol=1e-3, atol=2e-5, err_msg=f'{mod}.{k}')


@pytest.mark.parametrize('rnn', ['gilr', 'gru'])
def test_action_only_dx_equals_full_dx(rnn, monkeypatch):
    """Actor step: the critic's first layer forms only the action-encoding block of dX (frozen critic, detached embedding).
    Same seeds -> same logs and parameters as with the full dX GEMM (the state / embedding blocks were never consumed)."""
    from offpolicy_rnn import alg_init
    runs = []
    for flag in ('1', '0'):
        monkeypatch.setenv('RESEL_ACTION_ONLY_DX', flag)
        torch.manual_seed(5)
        np.random.seed(5)
        alg = alg_init(make_parameter(rnn, sac_batch_size=40))
        rs = np.random.RandomState(9)
        for n in [12, 5, 7, 12, 4, 9, 6]:
            o, a, r = _synth(rs, n, 5, 3)
            _push(alg.replay_buffer, o, a, r, early_done=(n != 12))
        torch.manual_seed(200)
        np.random.seed(200)
        logs = []
        for _ in range(3):
            logs.append(dict(alg.train_one_batch()))
            alg.grad_num += 1
        runs.append((logs, alg.policy.store.flat.detach().clone()))
    for a, b in zip(runs[0][0], runs[1][0]):
        for k in b:
            va, vb = (a[k][0] if isinstance(a[k], tuple) else a[k]), (b[k][0] if isinstance(b[k], tuple) else b[k])
            assert va == pytest.approx(vb, rel=1e-5, abs=1e-6), k
    np.testing.assert_allclose(runs[0][1].cpu(), runs[1][1].cpu(), rtol=1e-5, atol=1e-7)


def test_full_size_step_runs_and_is_finite():
    """BASELINE config-2 shapes (smamba_s32_c16_b2_nln, D=256, T=1024) at a reduced row count: finite, non-trivial update."""
    from offpolicy_rnn import alg_init
    from bench import build_trainer
    alg = build_trainer('smamba_s32_c16_b2_nln', B=4, T=1024, seed=0)
    before = alg.policy.store.flat.clone()
    log = alg.train_one_batch()
    assert log['real_batch_size'] == 4 * 1024 and log['real_batch_traj_num'] == 4
    for k, v in log.items():
        v = v[0] if isinstance(v, tuple) else v
        assert np.isfinite(v), k
    assert not torch.equal(before, alg.policy.store.flat)


def test_cgpt_layer_gpu_vs_oracle():
    """cgpt decoder (bf16 MFMA attention + bf16 projections) against the oracle restatement; north_star: 1e-2 for bf16.  Parity unpinned."""
    from offpolicy_rnn.models.rnn_base import RNNBase
    from offpolicy_rnn.models.flash_attention.TransformerFlashAttention import PackedSeqs
    from oracle import network as NW
    D, lid = 128, 'cgpt_h4_l2_p0.0_ml256_rms'
    torch.manual_seed(3)
    net = RNNBase(D, D, [], ['linear'], [lid])
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    x = torch.randn(3, 90, D)
    table = np.zeros((3, 90), dtype=np.int64)
    table[0, :3], table[1, :2], table[2, :1] = (1, 50, 30), (40, 45), (90,)
    w = torch.randn(3, 90, D)
    xr = x.clone().requires_grad_(True)
    pr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = NW.rnn_base_forward(pr, dict(layer_type=[lid], activation=['linear']), xr, NW.Flags(seqlens=torch.from_numpy(table)))
    (ref * w).sum().backward()
    net.to('cuda')
    net.eval()
    xg = x.clone().cuda().requires_grad_(True)
    hid = net.make_init_state(3, torch.device('cuda'))
    hid.set_attention_concat_mask(PackedSeqs(table, 90, torch.device('cuda')))
    y, _, _ = net.meta_forward(xg, hid)
    (y * w.cuda()).sum().backward()
    def rel(got, want, name, tol):
        e = (got.detach().cpu() - want.detach()).abs().max().item() / max(want.abs().max().item(), 1e-3)
        print(f'MEASURED cgpt small layer {name}: max err / max|ref| = {e:.3e} (bound {tol:g})')
        assert e < tol, (name, e)
    rel(y, ref, 'y', 1e-2)                        # measured 5.6e-4
    rel(xg.grad, xr.grad, 'dx', 2e-2)             # measured 1.2e-3
    for k, p in net.named_parameters():
        rel(p.grad, pr[k].grad, 'd ' + k, 2e-2)   # measured <= 3.6e-3


def test_cgpt_td3_update_gpu_vs_oracle():
    """BASELINE config-3 family (cgpt TD3) end to end at reduced width: one update on cuda:0 vs the oracle trainer (bf16 tolerances)."""
    from offpolicy_rnn import alg_init
    from oracle.trainer import OracleTrainer, default_parameter
    from test_oracle_golden import _push as opush
    lid, lens = 'cgpt_h2_l2_p0.0_ml64', [12, 5, 7, 12, 4, 9, 6]
    torch.manual_seed(5)
    alg = alg_init(make_parameter(lid, D=64, algo='td3', sac_batch_size=33))
    par = default_parameter(rnn=lid, D=64, algo='td3', sac_batch_size=33, policy_embedding_dim=16, value_embedding_dim=16,
                            policy_uni_model_input_mapping_dim=16, value_uni_model_input_mapping_dim=16, max_buffer_transition_num=5000)
    cpu_sd = lambda m: {k: {n: t.detach().cpu() for n, t in d.items()} for k, d in m.state_dict().items()}
    tr = OracleTrainer(par, 5, 3, 12, policy_state=cpu_sd(alg.policy), value_state=cpu_sd(alg.values[0]))
    rs = np.random.RandomState(9)
    for n in lens:
        o, a, r = _synth(rs, n, 5, 3)
        _push(alg.replay_buffer, o, a, r, early_done=(n != 12))
        opush(tr.buffer, o, a, r, early_done=(n != 12))
    logs = []
    for runner in (alg, tr):
        torch.manual_seed(200)
        np.random.seed(200)
        logs.append(runner.train_one_batch())
    for k, v in logs[1].items():
        got = logs[0][k][0] if isinstance(logs[0][k], tuple) else logs[0][k]
        want = v[0] if isinstance(v, tuple) else v
        assert got == pytest.approx(want, rel=5e-2, abs=5e-2), (k, got, want)


@pytest.mark.gpu
@pytest.mark.parametrize('rnn,ragged,algo,per,clip', [
    ('smamba_s8_c4_b1_nln', False, 'sac', 1, False), ('gilr', False, 'sac', 1, False), ('smamba_s8_c4_b1_nln', True, 'sac', 1, False),
    ('lru', False, 'td3', 1, False), ('cgpt_h1_l2_p0.0_ml64_rms', False, 'td3', 1, False),
    # the reference's published flag set (gen_tmuxp_mamba_pomdp.py:81): the actor steps on every second update - two graphs alternate
    ('smamba_s8_c4_b1_nln', False, 'sac', 2, False), ('gilr', False, 'td3', 2, False), ('smamba_s8_c4_b1_nln', True, 'sac', 2, False),
    # gradient clipping (reference :239-250, 274-287) at norms the gradients exceed, with and without the alternating actor step
    ('smamba_s8_c4_b1_nln', False, 'sac', 1, True), ('gilr', False, 'sac', 2, True), ('lru', False, 'sac', 1, 'value')])
def test_graphed_update_equals_the_eager_update(rnn, ragged, algo, per, clip, monkeypatch):
    """The whole update replayed from ONE hipGraph (algorithm/graphed_update.py: sampling plan, REDQ subset and AdamW step factors in
    static buffers refreshed before each replay) against the eager update: same seeds, actor noise off (the captured generator draws
    from graph-safe Philox offsets), four updates (eight with policy_update_per = 2) - logged scalars and every parameter to 2e-5."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    import numpy as np
    from test_host_logic import _push, _synth, make_parameter
    from offpolicy_rnn import alg_init
    from offpolicy_rnn.utility import rng
    from offpolicy_rnn.algorithm.graphed_update import GraphedUpdate
    monkeypatch.setattr(rng, 'randn', lambda shape, device, dtype=torch.float32: torch.zeros(tuple(shape), dtype=dtype, device=device))

    def build():
        torch.manual_seed(0)
        np.random.seed(0)
        extra = dict(policy_update_per=per)
        if clip == 'value':                                     # element-wise clipping of the embedding gradients (the torch spelling, in place)
            extra.update(value_embedding_max_gradnorm=1e-3, policy_embedding_max_gradnorm=1e-3)
        elif clip:
            extra.update(value_max_gradnorm=0.05, policy_max_gradnorm=0.01)
        alg = alg_init(make_parameter(rnn, algo=algo, sac_batch_size=4 * 12 - 1, cuda_inference=True, **extra))
        assert GraphedUpdate.refusal(alg) is None
        rs = np.random.RandomState(3)
        for i in range(8):                                      # equal lengths: one batch shape; ragged: the shape changes between updates
            n = (12, 9, 7, 12, 5, 12, 10, 8)[i] if ragged else 12
            o, a, r = _synth(rs, n, 5, 3)
            _push(alg.replay_buffer, o, a, r, early_done=(n != 12))
        np.random.seed(11)
        return alg

    def state(alg):
        return [alg.policy.store.flat.detach().clone(), alg.values[0].store.flat.detach().clone(), alg.target_values[0].store.flat.detach().clone(),
                alg.log_sac_alpha.detach().clone()]

    n_upd = (16 if ragged else 4) * per
    rs_new = np.random.RandomState(99)
    fresh = [_synth(rs_new, n, 5, 3) for n in (12, 6)]         # trajectories that enter the ring in the middle of the run

    def push_fresh(alg):
        for (o, a, r), n in zip(fresh, (12, 6)):
            _push(alg.replay_buffer, o, a, r, early_done=(n != 12))

    eager = build()
    logs_e = []
    for i in range(n_upd):
        if i == n_upd - 2:
            push_fresh(eager)
        logs_e.append(dict(eager.train_one_batch()))
        eager.grad_num += 1
    graphed = build()
    # every step() is ONE update: update 0 runs eagerly (warm-up), a batch shape is recorded on its second visit and replayed from
    # then on; ragged: at most two graphs live at a time (least recently used dropped), the other updates run eagerly
    st0 = np.random.get_state()[1].copy()
    g = GraphedUpdate(graphed, warmup=1, max_graphs=2 if ragged else 4)
    assert (np.random.get_state()[1] == st0).all()              # the constructor consumes no draw of the trainer's random streams
    logs_g = []
    for i in range(n_upd):
        if i == n_upd - 2:
            push_fresh(graphed)                                 # the ring mirror is refreshed outside the graph, in place
        logs_g.append(dict(g.step()))
        graphed.grad_num += 1
    torch.cuda.synchronize()
    print(f'graphs recorded {len(g.graphs)}, eager updates {g.eager_fallbacks} of {n_upd}')
    assert len(g.graphs) <= (2 if ragged else 4) and g.eager_fallbacks >= 1
    # equal lengths: warm-up + at most one first visit per launch sequence (the ring refill adds one); ragged shapes must recur to be recorded
    assert ragged or (g.graph is not None and g.eager_fallbacks <= 3 * per)
    if per == 2 and not ragged:
        assert {k[-1] for k in g.graphs} == {True, False}, 'one recording with and one without the actor step'
    if clip is True:                                            # the clip was active: the logged norm (pre-clip, reference :241) exceeds the bound
        assert logs_g[-2 if per == 2 else -1]['policy_grad_norm'] > 0.01 and logs_g[-1]['value_grad_norm'] > 0.05
    # not bit for bit: the entropy coefficient's torch AdamW runs in its `capturable` form (step count and bias corrections as
    # fp32 device tensors instead of Python floats), and the coefficient enters every loss
    # cgpt: the attention runs in bf16 - an operand that differs in its last fp32 bit (the replayed GEMMs scale their fp16 planes with
    # magnitude handles accumulated over replays, the eager ones with fresh ones) can round to the next bf16 value, 4e-3 of an element
    rtol, atol = (5e-3, 3e-4) if rnn.startswith('cgpt') else (2e-5, 2e-7 if n_upd <= 16 else 1e-6)     # 32 chained updates: last-bit drift of weights near zero
    for nm, a, b in zip(('policy', 'value', 'target value', 'log alpha'), state(graphed), state(eager)):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=rtol, atol=atol, err_msg=nm)
    for le, lg in zip(logs_e, logs_g):
        assert set(le) == set(lg)
        for k in le:
            ve = le[k][0] if isinstance(le[k], tuple) else le[k]
            vg = lg[k][0] if isinstance(lg[k], tuple) else lg[k]
            assert abs(ve - vg) <= max(rtol, 100 * atol if rnn.startswith('cgpt') else 0) * max(1.0, abs(ve)), (k, ve, vg)


@pytest.mark.gpu
def test_graphed_update_with_dropout_replays_what_its_eager_path_runs(monkeypatch):
    """cgpt with dropout 0.1 (configs[2]'s layer id at a small size) through GraphedUpdate: the masks of an update are keyed on a
    per-update host count baked into the kernel nodes + a device word that a node of the graph advances, so (a) a trainer stepped
    through replays and one stepped through the SAME object's eager path (warm-up never ends) agree update by update, and (b) two
    replays of one batch shape do not repeat each other's masks (the logged losses of consecutive updates differ)."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    import numpy as np
    from test_host_logic import _push, _synth, make_parameter
    from offpolicy_rnn import alg_init
    from offpolicy_rnn.utility import rng
    from offpolicy_rnn.algorithm.graphed_update import GraphedUpdate
    monkeypatch.setattr(rng, 'randn', lambda shape, device, dtype=torch.float32: torch.zeros(tuple(shape), dtype=dtype, device=device))
    monkeypatch.setattr(rng, 'randn_like', lambda t: torch.zeros_like(t))

    def build():
        torch.manual_seed(0)
        np.random.seed(0)
        alg = alg_init(make_parameter('cgpt_h1_l2_p0.1_ml64_rms', algo='td3', sac_batch_size=4 * 12 - 1, cuda_inference=True))
        rs = np.random.RandomState(3)
        for i in range(8):
            o, a, r = _synth(rs, 12, 5, 3)
            _push(alg.replay_buffer, o, a, r, early_done=False)
        np.random.seed(11)
        return alg

    runs = []
    for warmup in (1, 10 ** 6):
        alg = build()
        g = GraphedUpdate(alg, warmup=warmup)
        logs = []
        for _ in range(5):
            logs.append(dict(g.step()))
            alg.grad_num += 1
        torch.cuda.synchronize()
        runs.append((logs, alg.policy.store.flat.detach().clone(), alg.values[0].store.flat.detach().clone(), len(g.graphs), g.eager_fallbacks))
        g.close()
    (lg, pg, vg, ng, eg), (le, pe, ve, ne, ee) = runs
    assert ng == 1 and eg <= 3 and ne == 0 and ee == 5
    np.testing.assert_allclose(pg.cpu().numpy(), pe.cpu().numpy(), rtol=5e-3, atol=1e-4)          # bf16 attention: see the test above
    np.testing.assert_allclose(vg.cpu().numpy(), ve.cpu().numpy(), rtol=5e-3, atol=1e-4)
    val = lambda v: v[0] if isinstance(v, tuple) else v
    for a, b in zip(lg, le):
        for k in a:
            assert abs(val(a[k]) - val(b[k])) <= 1e-2 * max(1.0, abs(val(b[k]))), (k, a[k], b[k])
    assert len({round(val(l['critic_loss']), 7) for l in lg}) == len(lg)
