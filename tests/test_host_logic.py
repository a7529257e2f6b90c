"""Host-side logic of the product on CPU (no GPU): API surface, layer-id grammar, batch layout, state-dict layout and
the `train_one_batch` orchestration against the golden logs produced by the reference itself.

Kernels are replaced by the CPU oracle through the explicit `oracle_ops` fixture (tests/oracle_backend.py); the real
HIP kernels are checked on the GPU box by tests/test_hip_ops.py and tests/test_trainer_gpu.py."""
import ctypes
import json
import os
import re
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, PKG, ROOT, load_golden, nested

META = json.load(open(os.path.join(GOLDEN, 'train_meta.json')))


def make_parameter(rnn='gru', D=32, algo='sac', env='synthetic-o5-a3-T12', **over):
    sys.argv = ['test']
    from offpolicy_rnn import Parameter
    p = Parameter()
    p.alg_name = ('sac' if algo == 'sac' else 'td3') + '_rnn_full_horizon_redQ_sep_optim'
    p.env_name = env
    p.value_net_num = 1
    for w in ('value', 'policy'):
        setattr(p, f'{w}_embedding_layer_type', ['fc', rnn, 'fc'])
        setattr(p, f'{w}_embedding_activations', ['elu', 'elu', 'linear'])
        setattr(p, f'{w}_embedding_hidden_size', [D, D])
        setattr(p, f'{w}_hidden_size', [D, D])
        setattr(p, f'{w}_activations', ['elu', 'elu', 'linear'])
        setattr(p, f'{w}_embedding_dim', 16)
        setattr(p, f'{w}_uni_model_input_mapping_dim', 16)
    p.value_layer_type = ['efc-8'] * 3
    p.policy_layer_type = ['fc'] * 3
    p.state_action_encoder = True
    p.last_state_input = True
    p.alpha_lr = 1e-4
    p.policy_update_per = 1
    p.max_buffer_transition_num = 5000
    for k, v in over.items():
        setattr(p, k, v)
    return p


# ------------------------------------------------------------------------------------------------ API surface
def test_parameter_defaults_match_reference_flags():
    sys.argv = ['test']
    from offpolicy_rnn import Parameter
    p = Parameter()
    expect = dict(env_name='HalfCheetah-v2', alg_name='sac_mlp', seed=1, policy_lr=3e-4, rnn_policy_lr=1e-5, alpha_lr=1e-2,
                  value_lr=1e-3, rnn_value_lr=1e-4, value_net_num=2, utd=1, redq_m=2, gamma=0.99, sac_tau=0.995, sac_alpha=0.2,
                  target_entropy_ratio=1.5, sac_batch_size=1024, base_algorithm='sac', sample_std=0.1,
                  target_action_noise_std=0.04, target_action_noise_clip=0.12, policy_update_per=1, max_buffer_transition_num=1000000,
                  value_embedding_layer_type=['fc', 'gru', 'fc', 'fc'], policy_hidden_size=[256, 128], cuda_inference=False,
                  state_action_encoder=False, policy_max_gradnorm=None, valid_number_post_randomized=256)
    for k, v in expect.items():
        assert getattr(p, k) == v, k
    sys.argv = ['test', '--cuda_inference', '--value_layer_type', 'efc-8', 'efc-8', '--policy_uni_model_input_mapping_dim', 'auto',
                '--sac_batch_size', '1999']
    p = Parameter()
    assert p.cuda_inference is True and p.value_layer_type == ['efc-8', 'efc-8'] and p.policy_uni_model_input_mapping_dim == 'auto'
    assert p.sac_batch_size == 1999


def test_alg_init_table(oracle_ops):
    from offpolicy_rnn import alg_init
    from offpolicy_rnn.algorithm.sac_full_length_rnn_redq_sep_optim import SACFullLengthRNNREDQ_SEP_OPTIM
    from offpolicy_rnn.algorithm.td3_full_length_rnn_redq_sep_optim import TD3FullLengthRNNREDQ_SEP_OPTIM
    alg = alg_init(make_parameter())
    assert isinstance(alg, SACFullLengthRNNREDQ_SEP_OPTIM)
    for attr in ('policy', 'values', 'target_values', 'replay_buffer', 'log_sac_alpha', 'grad_num', 'parameter', 'train',
                 'train_one_batch', 'save', 'load', 'step'):
        assert hasattr(alg, attr), attr
    p = make_parameter(algo='td3')
    alg = alg_init(p)
    assert isinstance(alg, TD3FullLengthRNNREDQ_SEP_OPTIM) and p.base_algorithm == 'td3' and p.no_alpha_auto_tune is True
    assert alg.log_sac_alpha.item() == 0.0
    with pytest.raises(NotImplementedError):
        alg_init(make_parameter(alg_name='sac_mlp'))
    with pytest.raises(NotImplementedError):
        alg_init(make_parameter(alg_name='no_such_alg'))
    with pytest.raises(AssertionError):                       # critic MLP must be an ensemble stack
        alg_init(make_parameter(value_layer_type=['fc'] * 3))


def test_layer_id_grammar_and_sizes():
    from offpolicy_rnn.models.rnn_base import RNNBase, parse_smamba_id
    table = json.load(open(os.path.join(GOLDEN, 'layer_ids.json')))
    for lid, e in table.items():
        net = RNNBase(32, 32, [], ['linear'], [lid])
        assert net.rnn_hidden_state_input_size[0] == e['hidden'], lid
        assert sum(p.numel() for p in net.parameters()) == e['nparam'], lid
        if lid.startswith('smamba'):
            c = parse_smamba_id(lid)
            for k in ('d_conv', 'd_state', 'block_num', 'rms_norm', 'use_ff'):
                assert c[k] == e[k], (lid, k)
    with pytest.raises(AssertionError):
        RNNBase(16, 32, [], ['linear'], ['smamba'])             # smamba needs in == out
    for lid in ('gpt_h8_l6', 'mamba_s16', 'lstm', 'conv1d_4'):
        with pytest.raises(NotImplementedError):
            RNNBase(32, 32, [], ['linear'], [lid])


@pytest.mark.parametrize('name', list(META))
def test_state_dict_layout_equals_reference(name, oracle_ops):
    """Module order, parameter names and shapes of actor / critic equal the reference's (checkpoint compatibility)."""
    from offpolicy_rnn import alg_init
    m = META[name]
    g = load_golden(f'train_{name}.npz')
    alg = alg_init(make_parameter(m['rnn'], algo=m['algo']))
    for prefix, model in (('policy0|', alg.policy), ('value0|', alg.values[0])):
        ref = nested(g, prefix)
        sd = model.state_dict()
        assert list(sd.keys()) == list(ref.keys()), prefix
        for mod in ref:
            assert {k: tuple(v.shape) for k, v in sd[mod].items()} == {k: tuple(v.shape) for k, v in ref[mod].items()}, (prefix, mod)
        model.load_state_dict(ref)                              # and it loads


def test_cgpt_layer_id_and_module_equals_oracle_restatement(oracle_ops):
    """cgpt: layer-id grammar, parameter layout, and the block structure against the oracle restatement (attention core =
    oracle on both sides here; the HIP attention kernels are checked on the GPU box).  flash_attn is absent from the reference
    checkout, so there is no golden vector: parity unpinned."""
    from offpolicy_rnn.models.rnn_base import RNNBase, parse_cgpt_id
    from offpolicy_rnn.models.flash_attention.TransformerFlashAttention import PackedSeqs
    from oracle import network as NW
    assert parse_cgpt_id('cgpt_h8_l6_p0.1_ml1024_rms') == dict(nhead=8, nlayer=6, pdrop=0.1, maxlength=1024, ln=False)
    assert parse_cgpt_id('cgpt') == dict(nhead=8, nlayer=4, pdrop=0.1, maxlength=1024, ln=True)
    D = 64
    for lid, ln in (('cgpt_h2_l2_p0.0_ml64', True), ('cgpt_h4_l1_p0.0_rms', False)):
        torch.manual_seed(1)
        net = RNNBase(D, D, [], ['linear'], [lid])
        cfg = parse_cgpt_id(lid)
        per_layer = 3 * D * D + 3 * D + D * D + D + 4 * D * D + 4 * D + 4 * D * D + D + 2 * (2 * D if ln else D)
        assert sum(p.numel() for p in net.parameters()) == cfg['nlayer'] * per_layer + (2 * D if ln else D) + D * D + D
        assert net.rnn_hidden_state_input_size == [cfg['maxlength']]
        x = torch.randn(2, 11, D)
        table = np.array([[1, 6, 3], [4, 7, 0]])
        hid = net.make_init_state(2, torch.device('cpu'))
        hid.set_attention_concat_mask(PackedSeqs(table, 11, torch.device('cpu')))
        net.eval()
        y, _, _ = net.meta_forward(x, hid)
        sd = {k: v.detach() for k, v in net.state_dict().items()}
        padded = np.zeros((2, 11), dtype=np.int64)                   # the reference pads the table to the row length (:360-361)
        padded[:, :3] = table
        ref = NW.rnn_base_forward(sd, dict(layer_type=[lid], activation=['linear']), x, NW.Flags(seqlens=torch.from_numpy(padded)))
        np.testing.assert_allclose(y.detach(), ref, rtol=2e-2, atol=2e-2)           # bf16 casts on both sides
        assert torch.all(y[0, 10] == 0)                                              # padding slot outside every sequence


# ------------------------------------------------------------------------------------------------ batch layout
def _push(buf, o, a, r, early_done):
    from offpolicy_rnn.buffers.transition_buffer.replay_memory import Transition
    n = len(a)
    for t in range(n):
        buf.mem_push(Transition(
            state=o[t:t + 1], last_state=o[t - 1:t] if t > 0 else np.zeros((1, o.shape[1])),
            last_action=a[t - 1:t] if t > 0 else np.zeros((1, a.shape[1])), action=a[t:t + 1], next_state=o[t + 1:t + 2],
            reward=float(r[t]), logp=None, mask=1, start=(t == 0), done=(t == n - 1),
            reward_input=np.array([[r[t - 1] if t > 0 else 0.0]]), timeout=(t == n - 1) and not early_done))


def _synth(rs, n, obs, act):
    return rs.randn(n + 1, obs), np.tanh(rs.randn(n, act)), rs.randn(n)


@pytest.mark.parametrize('case', ['skip2_nonest', 'skip18_nest', 'skip2_nest_rmask', 'skip2_fixedT'])
def test_sample_trajs_equals_reference(case):
    from offpolicy_rnn.buffers.transition_buffer.nested_replay_memory import NestedMemoryArray
    g = load_golden('sample_trajs.npz')
    cfg = g[f'{case}|cfg']
    hist, nest, rmask, bs, maxT = [int(v) for v in cfg[:5]]
    buf = NestedMemoryArray(1000, maxT, additional_history_len=hist)
    rs = np.random.RandomState(7)
    for n in [int(v) for v in cfg[5:]]:
        o, a, r = _synth(rs, n, 4, 2)
        _push(buf, o, a, r, early_done=(n != 12))
    np.random.seed(123)
    res, total, valid, table = buf.sample_trajs(bs, None, randomize_mask=bool(rmask), valid_number_post_randomized=9,
                                                equalize_data_of_each_traj=True, nest_stack_trajs=bool(nest))
    assert total == int(g[f'{case}|total'])
    np.testing.assert_array_equal(valid, g[f'{case}|valid'].astype(np.float32))
    np.testing.assert_array_equal(table, g[f'{case}|table'])
    for f in res._fields:
        if getattr(res, f) is not None:
            np.testing.assert_array_equal(getattr(res, f), g[f'{case}|{f}'].astype(np.float32), err_msg=f)
    assert res.logp is None


def test_buffer_eviction_and_edge_cases():
    from offpolicy_rnn.buffers.transition_buffer.nested_replay_memory import NestedMemoryArray
    buf = NestedMemoryArray(30, 12, additional_history_len=1)
    rs = np.random.RandomState(0)
    for n in (12, 12, 12, 5):
        o, a, r = _synth(rs, n, 3, 2)
        _push(buf, o, a, r, early_done=(n != 12))
    assert len(buf) == 3 and buf.size == 29                    # the oldest trajectory was evicted (30-transition cap)
    np.random.seed(0)
    res, total, valid, table = buf.sample_trajs(29, None, equalize_data_of_each_traj=True, nest_stack_trajs=True)
    assert total == 29 and res.state.shape[0] == 3             # 14 / 14 / 7 slots: no two fit one 16-slot row
    # a single-step trajectory still gets its pre-step slot
    buf2 = NestedMemoryArray(100, 12, additional_history_len=1)
    o, a, r = _synth(rs, 1, 3, 2)
    _push(buf2, o, a, r, early_done=True)
    res, total, valid, table = buf2.sample_trajs(1, None, equalize_data_of_each_traj=True)
    assert total == 1 and res.state.shape[1] == 4 and list(res.start[0, :, 0]) == [1, 1, 1, 1] and list(valid[0, :, 0]) == [0, 0, 1, 0]


# ------------------------------------------------------------------------------------------------ trainer orchestration
def _trainer(name, oracle_ops):
    from offpolicy_rnn import alg_init
    m = META[name]
    g = load_golden(f'train_{name}.npz')
    torch.manual_seed(0)
    alg = alg_init(make_parameter(m['rnn'], algo=m['algo'], sac_batch_size=m['sac_batch_size']))
    alg.policy.load_state_dict(nested(g, 'policy0|'))
    alg.values[0].load_state_dict(nested(g, 'value0|'))
    alg._value_update(tau=0.0)
    rs = np.random.RandomState(9)
    for n in m['lens']:
        o, a, r = _synth(rs, n, 5, 3)
        _push(alg.replay_buffer, o, a, r, early_done=(n != 12))
    return alg, g, m


@pytest.mark.parametrize('name', ['gru_sac', 'gru_td3', 'gilr_sac', 'lru_sac'])
def test_train_one_batch_equals_reference_logs(name, oracle_ops):
    """Three consecutive updates with the reference's seeds: every entry of the returned dict and every parameter."""
    alg, g, m = _trainer(name, oracle_ops)
    assert alg._get_skip_len() == m['skip_len'] and alg.allow_nest_stack == m['nest']
    torch.manual_seed(200)
    np.random.seed(200)
    for it in range(3):
        log = alg.train_one_batch()
        alg.grad_num += 1
        ref = m['logs'][it]
        assert set(ref) <= set(log), set(ref) - set(log)
        for k, v in ref.items():
            got = log[k][0] if isinstance(log[k], tuple) else log[k]
            assert got == pytest.approx(v, rel=2e-3, abs=2e-4), (it, k, got, v)
    for pre, net in (('policy3|', alg.policy), ('value3|', alg.values[0]), ('target3|', alg.target_values[0])):
        sd = net.state_dict()
        for mod, d in nested(g, pre).items():
            for k, v in d.items():
                np.testing.assert_allclose(sd[mod][k].detach(), v, rtol=2e-3, atol=2e-5, err_msg=f'{pre}{mod}.{k}')
    np.testing.assert_allclose(alg.log_sac_alpha.detach(), g['log_alpha3'], rtol=1e-5, atol=1e-6)


def test_train_one_batch_smamba_equals_oracle_gpu_semantics(oracle_ops):
    """smamba honours start / mask on the training path (the reference only does so on CUDA tensors), so the yardstick is
    the oracle trainer in 'gpu' semantics - itself pinned to the reference layer-by-layer in test_oracle_golden.py."""
    from oracle.trainer import OracleTrainer, default_parameter
    alg, g, m = _trainer('smamba_sac', oracle_ops)
    par = default_parameter(rnn=m['rnn'], D=32, algo='sac', sac_batch_size=m['sac_batch_size'], policy_embedding_dim=16,
                            value_embedding_dim=16, policy_uni_model_input_mapping_dim=16, value_uni_model_input_mapping_dim=16,
                            max_buffer_transition_num=5000)
    tr = OracleTrainer(par, 5, 3, 12, smamba_semantics='gpu', policy_state=nested(g, 'policy0|'), value_state=nested(g, 'value0|'))
    rs = np.random.RandomState(9)
    from test_oracle_golden import _push as opush
    for n in m['lens']:
        o, a, r = _synth(rs, n, 5, 3)
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
    for a, b in zip(*logs):
        for k, v in b.items():
            got = a[k][0] if isinstance(a[k], tuple) else a[k]
            want = v[0] if isinstance(v, tuple) else v
            assert got == pytest.approx(want, rel=2e-3, abs=2e-4), k
    sd = alg.policy.state_dict()
    for mod, d in tr.policy.items():
        for k, v in d.items():
            np.testing.assert_allclose(sd[mod][k].detach(), v.detach(), rtol=2e-3, atol=2e-5, err_msg=f'{mod}.{k}')


def test_flat_parameter_store_views_and_checkpoint_roundtrip(oracle_ops, tmp_path):
    from offpolicy_rnn import alg_init
    alg = alg_init(make_parameter('gilr'))
    st = alg.policy.store
    assert st.flat.numel() == st.numel + st.extra
    for p, o, n in st.slices:
        assert p.data_ptr() == st.flat.data_ptr() + 4 * o and p.grad.data_ptr() == st.grad.data_ptr() + 4 * o and o % 4 == 0
    before = st.flat.clone()
    alg.save(str(tmp_path))
    names = sorted(os.listdir(tmp_path))
    assert 'log_sac_alpha.pt' in names and 'ContextualSACPolicy-0-embedding_model.pt' in names
    assert 'ContextualSACValue-0-target-universal_model.pt' in names
    with torch.no_grad():
        st.flat.add_(1.0)
    alg.load(str(tmp_path))
    for p, o, n in alg.policy.store.slices:                      # loaded in place: parameters are still views of the flat buffer
        assert p.data_ptr() == alg.policy.store.flat.data_ptr() + 4 * o
        assert torch.equal(alg.policy.store.flat[o:o + n], before[o:o + n])


# ------------------------------------------------------------------------------------------------ boundary / layout rules
def test_library_exports_every_header_symbol():
    header = open(os.path.join(ROOT, 'include', 'resel_hip.h')).read()
    declared = set(re.findall(r'\b(resel_[a-z0-9_]+)\s*\(', header))
    from offpolicy_rnn.hip import _lib
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip('library not built (run __graft_entry__.build())')
    h = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(h, name), name
    h.resel_abi_version.restype = ctypes.c_int
    assert h.resel_abi_version() == _lib.ABI_VERSION


def test_product_never_touches_the_oracle_and_has_no_cpu_fallback():
    bad = []
    for base, _, files in os.walk(PKG):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                txt = open(os.path.join(base, f)).read()
                if re.search(r'^\s*(from|import)\s+oracle\b', txt, re.M) or 'oracle_backend' in txt:
                    bad.append(os.path.join(base, f))
    assert not bad, bad
    from offpolicy_rnn.hip import ops
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops.gilr_scan(torch.zeros(1, 4, 8), torch.zeros(1, 4, 8))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops.selective_scan_tm(torch.zeros(1, 4, 8), torch.zeros(1, 4, 8), -torch.ones(8, 4), torch.zeros(1, 4, 4), torch.zeros(1, 4, 4))


# ------------------------------------------------------------------------------------------ host utilities added for the GPU path
def test_deferred_log_behaves_like_the_reference_dict():
    """train_one_batch() returns a dict (reference :430-467); DeferredLog keeps that contract while the device scalars are in
    flight: host entries readable at once, everything else materialises on first access, actor_loss stays a 1-tuple."""
    import torch
    from offpolicy_rnn.algorithm.sac_full_length_rnn_ensembleQ import DeferredLog
    log = DeferredLog(['critic_loss', 'actor_loss'], torch.tensor([1.5, -2.0]), pinned=False)
    log.set_host({'real_batch_size': 7})
    assert isinstance(log, dict) and log['real_batch_size'] == 7
    assert log['critic_loss'] == 1.5 and log['actor_loss'] == (-2.0,)
    assert set(log.keys()) == {'critic_loss', 'actor_loss', 'real_batch_size'} and len(log) == 3
    assert dict(**log)['critic_loss'] == 1.5 and 'critic_loss' in log and log.get('missing', 3) == 3


def test_subset_table_returns_the_requested_indices(monkeypatch):
    import oracle_backend
    oracle_backend.install(monkeypatch)
    from bench import build_trainer
    alg = build_trainer('gru', 2, 16)
    for sub in ([3, 5], [5, 3], [0, 7], list(range(8))):
        got = alg._subset_on_device(np.asarray(sub), 8)
        assert got.dtype == torch.int32 and got.tolist() == sub
    assert alg._subset_on_device(np.asarray([1, 2]), 8).data_ptr() == alg._subset_on_device(np.asarray([1, 2]), 8).data_ptr()


def test_gemm_table_is_well_formed_and_inert_without_a_gpu():
    """hip/gemm_tuning/gfx950.csv: validator header + one solution per (op, shape); the loader is a no-op on a CPU-only host."""
    from offpolicy_rnn.hip import gemm_select
    lines = [ln.strip() for ln in open(gemm_select.TABLE) if ln.strip()]
    assert [ln.split(',')[1] for ln in lines[:5]] == ['PT_VERSION', 'HIP_VERSION', 'HIPBLASLT_VERSION', 'GCN_ARCH_NAME', 'ROCBLAS_VERSION']
    keys = [tuple(ln.split(',')[:2]) for ln in lines[5:]]
    assert len(keys) == len(set(keys)) > 100 and all(len(ln.split(',')) == 4 for ln in lines[5:])
    assert any('gfx950' in ln for ln in lines[:5])
    if not torch.cuda.is_available():
        assert gemm_select.enable_tuned_gemms() is False
