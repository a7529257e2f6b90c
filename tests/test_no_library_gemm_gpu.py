"""Verdict r05 item 7, enforced: NO product of the path is left to the vendor GEMM library - neither in an update (whole trajectories: the
matrix-core editions of `resel_gemm_f32x` / `resel_gemm_bf16`; shapes they cannot read: csrc/gemm_any.hip) nor in a rollout step (one token
per environment: the rows form of gemm_any.hip).  The torch profiler watches the ATen dispatcher while the product runs: no aten::mm / addmm /
bmm / baddbmm / linear / matmul / einsum may appear, at full width and at the odd small sizes of the unit tests alike (reference call sites:
models/ensemble_linear_model.py:36-49, models/rnn_base.py:131-136 `fc`, smamba/mamba.py:257-305, flash_attention/TransformerFlashAttention.py:64-121)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
LIB = ('aten::mm', 'aten::addmm', 'aten::bmm', 'aten::baddbmm', 'aten::linear', 'aten::matmul', 'aten::einsum', 'aten::addmv', 'aten::mv', 'aten::dot')


def _library_gemms(fn):
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        fn()
        torch.cuda.synchronize()
    return sorted({e.name for e in prof.events() if e.name in LIB})


@pytest.mark.parametrize('rnn,algo', [('smamba_s8_c4_b1_nln', 'sac'), ('gilr', 'td3'), ('lru', 'sac'), ('gru', 'sac'), ('cgpt_h1_l2_p0.1_ml64_rms', 'td3'),
                                      ('gilr_lstm', 'sac'), ('conv1d_4', 'sac'), ('mamba_s8_c3', 'sac')])
def test_an_update_issues_no_library_gemm(rnn, algo):
    """Small odd sizes (47 tokens, 5-wide observations, 3-wide actions, D = 32): every shape rule of the matrix-core editions is violated
    somewhere - rows of 12 and 20 bytes, reductions of 3, a rank-2 dt_proj."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from test_host_logic import _push, _synth, make_parameter
    from offpolicy_rnn import alg_init
    torch.manual_seed(0)
    np.random.seed(0)
    alg = alg_init(make_parameter(rnn, algo=algo, sac_batch_size=47, cuda_inference=True))
    rs = np.random.RandomState(3)
    for n in (12, 9, 7, 12, 5, 12):
        o, a, r = _synth(rs, n, 5, 3)
        _push(alg.replay_buffer, o, a, r, early_done=(n != 12))
    alg.train_one_batch()
    alg.grad_num += 1

    def update():
        log = dict(alg.train_one_batch())
        assert all(np.isfinite(v[0] if isinstance(v, tuple) else v) for v in log.values())
    assert _library_gemms(update) == []


@pytest.mark.parametrize('rnn,algo,rows', [('smamba_s32_c16_b2_nln', 'sac', 8), ('cgpt_h8_l2_p0.1_ml1024_rms', 'td3', 4)])
def test_a_full_width_update_issues_no_library_gemm(rnn, algo, rows):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from bench import build_trainer
    alg = build_trainer(rnn, rows, 1024, algo=algo)
    alg.train_one_batch()
    alg.grad_num += 1
    assert _library_gemms(lambda: alg.train_one_batch()) == []


@pytest.mark.parametrize('rnn', ['smamba_s32_c16_b2_nln', 'gilr', 'lru', 'gru', 'cgpt_h8_l2_p0.0_ml64_rms', 'mamba_s8_c3', 'conv1d_4', 'gilr_lstm'])
def test_a_rollout_step_issues_no_library_gemm(rnn):
    """The per-environment-step policy forward (T = 1), launched eagerly so that the profiler sees every op (the trainer's loop replays the same
    launches from a hipGraph)."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from bench import make_parameter, OBS, ACT
    from offpolicy_rnn import alg_init
    torch.manual_seed(0)
    alg = alg_init(make_parameter(rnn, 2, 64))
    alg.graph_step = None
    rs = np.random.RandomState(0)
    alg.state_np, alg.last_state_np, alg.last_action_np, alg.reward_np = rs.randn(1, OBS), rs.randn(1, OBS), np.tanh(rs.randn(1, ACT)), rs.randn(1, 1)
    alg.sample_hidden = alg._init_sample_hidden()
    alg.sample_action()

    def step():
        a = alg.sample_action()
        assert np.isfinite(a).all()
    assert _library_gemms(step) == []
