"""Value-level parity at BASELINE.json's OWN sizes.  Row independence of every kernel on the path makes this cheap: the
product runs the published embedding tower (fc -> sequence layer -> fc, D = 256) on a batch at the full row length, the
oracle restates the first two rows; outputs, dx and every parameter gradient of those rows are compared at north_star's fp32
bar (1e-4, relative to the largest reference magnitude of the compared tensor; the loss ignores the other rows).

  configs[1] / [3]: smamba_s32_c16_b2_nln, T' = 1043 (T = 1024 + skip 18 + 1), d_inner 512, N 32, conv K 16 - at B = 4 (32 workgroups: the
    library cuts the rows into time segments) AND at the bench's own B = 64 (512 workgroups: the one-pass forward + 8-step-checkpoint
    backward that `bench.py` times);
  configs[4]: gilr and lru, T' = 2003 (T = 2000 full episode), B = 16;
  gru at configs[1]'s B, T (T' = 1027, H = 256: 1026 sequential steps) and at configs[0]'s shape (B = 8, T = 128 -> T' = 130).
Flags as the trainer builds them: the pre-step slots of a row are `start`, validity covers the trajectory, one mid-row reset."""
import pytest
import torch

from oracle import network as NW

pytestmark = pytest.mark.gpu


def _close(got, ref, tol, name):
    got, ref = got.detach().float().cpu(), ref.detach().float()
    scale = max(ref.abs().max().item(), 1e-6)
    err = (got - ref).abs().max().item()
    assert torch.isfinite(got).all() and err <= tol * scale, f'{name}: max err {err:.3e} vs scale {scale:.3e}'


def _close_elementwise(got, ref, rtol, name, floor=1e-5):
    """north_star's rtol read element by element: |got - ref| <= rtol * |ref| + floor * max|ref| for EVERY element
    (the floor term only keeps elements that are zero by cancellation from being held to their own size)."""
    got, ref = got.detach().float().cpu(), ref.detach().float()
    bound = rtol * ref.abs() + floor * max(ref.abs().max().item(), 1e-6)
    excess = ((got - ref).abs() - bound).max().item()
    worst = ((got - ref).abs() / bound).max().item()
    assert torch.isfinite(got).all() and excess <= 0, f'{name}: worst element at {worst:.2f}x its bound (rtol {rtol}, floor {floor})'


@pytest.mark.parametrize('lid,B,L,skip', [('smamba_s32_c16_b2_nln', 4, 1043, 18), ('smamba_s32_c16_b2_nln', 64, 1043, 18), ('gilr', 16, 2003, 2), ('lru', 16, 2003, 2),
                                          ('gru', 4, 1027, 2), ('gru', 8, 130, 2)])
def test_embedding_tower_at_the_baseline_row_length_vs_oracle(lid, B, L, skip):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from offpolicy_rnn.models.rnn_base import RNNBase
    D, R = 256, 2
    torch.manual_seed(7)
    net = RNNBase(384, 128, [D, D], ['elu', 'elu', 'linear'], ['fc', lid, 'fc'])       # gen_tmuxp_mamba_pomdp.py:43-86 embedding tower
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, L, 384, generator=g) * 0.5
    start = torch.zeros(B, L, 1)
    start[:, :skip] = 1                                    # pre-step slots
    start[0, min(700, 2 * L // 3)] = 1                     # a packed second trajectory in row 0
    mask = torch.ones(B, L, 1)
    mask[:, :skip - 1] = 0
    mask[:, -1] = 0
    w = torch.randn(B, L, 128, generator=g)
    w[R:] = 0                                              # the loss sees the first R rows only
    spec = dict(layer_type=['fc', lid, 'fc'], activation=['elu', 'elu', 'linear'])
    xr = x[:R].clone().requires_grad_(True)
    pr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    # gru: the oracle is the reference's own layer, torch.nn.GRU on the CPU (ATen `gru`, rnn_base.py:59,247) - north_star's
    # "reference CPU GRU path"; 1026 sequential steps at H = 256 (configs[1]'s row length for gru) and configs[0]'s B = 8, T = 128
    ref = NW.rnn_base_forward(pr, spec, xr, NW.Flags(rnn_start=start[:R], mask=mask[:R]), smamba_semantics='gpu',
                              gru_impl='aten')
    (ref * w[:R]).sum().backward()
    net.to('cuda')
    net.train()
    xg = x.cuda().requires_grad_(True)
    hid = net.make_init_state(B, torch.device('cuda'))
    hid.set_rnn_start(start.cuda())
    hid.set_mask(mask.cuda())
    y = net.meta_forward(xg, hid)[0]
    (y * w.cuda()).sum().backward()
    _close(y[:R], ref, 1e-4, f'{lid} y')
    _close_elementwise(y[:R], ref, 1e-4, f'{lid} y (element-wise)')
    _close(xg.grad[:R], xr.grad, 2e-4, f'{lid} dx')
    assert xg.grad[R:].abs().max().item() == 0.0           # rows are independent
    for k, p in net.named_parameters():
        g_ref = pr[k].grad
        if p.grad is None or g_ref is None:                 # a parameter the layer never reads (placeholders of the reference layout)
            assert (p.grad is None or p.grad.abs().max().item() == 0.0) and (g_ref is None or g_ref.abs().max().item() == 0.0), (lid, k)
            continue
        assert (p.grad.cpu() - g_ref).abs().max().item() <= 2e-4 * max(g_ref.abs().max().item(), 1e-6), (lid, k)
