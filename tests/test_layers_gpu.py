"""Sequence layers of the product on cuda:0 against vectors recorded from the reference's own modules
(tests/golden/layers.npz: forward, input gradient and every parameter gradient, rows with mid-row resets and masked
slots).  smamba ids are held to the reference's GPU-path semantics (`y_seq`: resets + conv input mask)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
IDS = ['gru', 'gilr', 'lru', 'gilr_lstm', 'conv1d_5', 'mamba_s8_c3', 'mamba_s4_c5_noff', 'smamba_s8_c6_b2_nln', 'smamba_s16_c4_b1',
       'smamba_s8_c5_b1_ff']


def T(a):
    return torch.from_numpy(np.array(a)).float()


@pytest.mark.parametrize('lid', IDS)
def test_layer_fwd_bwd_vs_reference_recording(lid):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from offpolicy_rnn.models.rnn_base import RNNBase
    g = load_golden('layers.npz')
    net = RNNBase(32, 32, [], ['linear'], [lid])
    pre = f'{lid}|p|'
    net.load_state_dict({k[len(pre):]: T(v) for k, v in g.items() if k.startswith(pre)})
    net.cuda()
    x = T(g['x']).cuda().requires_grad_(True)
    hid = net.make_init_state(x.shape[0], x.device)
    hid.set_rnn_start(T(g['start']).cuda())
    hid.set_mask(T(g['mask']).cuda())
    y, _, _ = net.meta_forward(x, hid)
    sm = lid.startswith('smamba')
    np.testing.assert_allclose(y.detach().cpu(), g[f'{lid}|y_seq' if sm else f'{lid}|y'], rtol=1e-4, atol=2e-5)
    (y * T(g['w']).cuda()).sum().backward()
    np.testing.assert_allclose(x.grad.cpu(), g[f'{lid}|dx_seq' if sm else f'{lid}|dx'], rtol=1e-3, atol=1e-4)
    tag = f'{lid}|gseq|' if sm else f'{lid}|g|'
    params = dict(net.named_parameters())
    checked = 0
    for k, v in g.items():
        if k.startswith(tag):
            name = k[len(tag):]
            np.testing.assert_allclose(params[name].grad.cpu(), v, rtol=2e-3, atol=3e-4, err_msg=name)
            checked += 1
    assert checked > 0


@pytest.mark.parametrize('lid', ['smamba_s16_c4_b1_nln', 'gilr', 'lru', 'cgpt_h8_l2_p0.0_ml1200_rms'])
def test_training_pass_fuses_the_elu_behind_smamba_without_changing_values(lid):
    """Inside `rnn_base.training_pass()` (the trainers' updates) the plain ELU behind a smamba / gilr / lru layer is applied by the layer's last
    kernel (the head GEMM's epilogue; the closing add + LayerNorm of the feed-forward block; cgpt's `output_fc` GEMM) and the layer's entry in the returned full-hidden
    record is None; outside, the record holds the PRE-activation sequence as in the reference (rnn_base.py:456-460).  Outputs and every
    gradient agree between the two forms."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from offpolicy_rnn.models.rnn_base import RNNBase, training_pass
    torch.manual_seed(3)
    net = RNNBase(96, 64, [256, 256], ['elu', 'elu', 'linear'], ['fc', lid, 'fc']).cuda()
    B, L = 4, 1100                                              # 4 400 tokens: the hand-written GEMMs (and their ELU epilogue) run
    x = torch.randn(B, L, 96, device='cuda')
    start = torch.zeros(B, L, 1, device='cuda')
    start[:, :5] = 1
    mask = torch.ones(B, L, 1, device='cuda')
    w = torch.randn(B, L, 64, device='cuda')
    res = []
    for fused in (False, True):
        net.zero_grad()
        xs = x.clone().requires_grad_(True)
        hid = net.make_init_state(B, torch.device('cuda'))
        hid.set_rnn_start(start)
        hid.set_mask(mask)
        if fused:
            with training_pass():
                y, _, full = net.meta_forward(xs, hid, require_full_hidden=True)
        else:
            y, _, full = net.meta_forward(xs, hid, require_full_hidden=True)
        (y * w).sum().backward()
        res.append((y.detach(), xs.grad, [None if p.grad is None else p.grad.clone() for p in net.parameters()], full))     # gilr holds an unused LayerNorm, as upstream
    (y0, g0, p0, f0), (y1, g1, p1, f1) = res
    assert f1[0] is None and torch.is_tensor(f0[0]) and f0[0].shape == (B, L, 256)
    assert (f0[0] < -1.0).any(), 'the unfused record is the pre-activation sequence (an ELU output never goes below -1)'
    rt = 1e-2 if lid.startswith('cgpt') else 2e-5          # cgpt's backward runs in bf16: a last-bit change of elu' moves roundings
    tol = lambda a, b: (a - b).abs().max().item() <= rt * max(b.abs().max().item(), 1e-6)
    assert tol(y1, y0) and tol(g1, g0)
    for a, b in zip(p1, p0):
        assert (a is None) == (b is None) and (a is None or tol(a, b))
