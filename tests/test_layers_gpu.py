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
