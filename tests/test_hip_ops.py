"""Parity of every C-ABI kernel (through the ctypes binding) against the CPU oracle.  GPU box only.

Tolerances: fp32 kernels against an fp32 CPU restatement of the same arithmetic - rtol 1e-4 (north_star's fp32 bar),
with an absolute floor that scales with the magnitude of the compared tensor (reductions over thousands of terms).
Read precisely: `close()` is NORM-WISE - max |got - ref| <= (rtol + atol_scale) * max |ref| over the whole tensor - and is what
gradients are held to; forward OUTPUTS additionally pass `close_fwd()`, the element-wise reading of north_star's rtol:
|got - ref| <= 1e-4 |ref| + 1e-5 max|ref| for every element.  Trainer-level tests
(tests/test_trainer_gpu.py) hold logged scalars to 2e-3 and parameters to 1e-3 / 2e-5 after three chained optimizer steps.
"""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import kernels as K

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from offpolicy_rnn.hip import ops as o
    return o


def close(got, ref, rtol=1e-4, atol_scale=2e-5, name=''):
    got = got.detach().float().cpu()
    ref = ref.detach().float()
    scale = max(ref.abs().max().item(), 1e-6)
    err = (got - ref).abs().max().item()
    assert torch.isfinite(got).all(), f'{name}: non-finite output'
    assert err <= rtol * scale + atol_scale * scale, f'{name}: max err {err:.3e} vs scale {scale:.3e}'


def close_fwd(got, ref, rtol=1e-4, floor=1e-5, name=''):
    """Forward outputs: north_star's rtol read ELEMENT-WISE - |got - ref| <= rtol * |ref| + floor * max|ref| for every element
    (the floor keeps elements that vanish by cancellation from being held to their own size) - on top of the norm-wise bound."""
    close(got, ref, rtol=rtol, name=name)
    got = got.detach().float().cpu()
    ref = ref.detach().float()
    bound = rtol * ref.abs() + floor * max(ref.abs().max().item(), 1e-6)
    worst = ((got - ref).abs() / bound).max().item()
    assert worst <= 1.0, f'{name}: worst element at {worst:.2f}x its element-wise bound (rtol {rtol}, floor {floor})'


def rnd(*shape, g=None, scale=1.0):
    return torch.randn(*shape, generator=g) * scale


def make_start(B, L, g, p=0.03):
    s = (torch.rand(B, L, generator=g) < p).float()
    s[:, 0] = 1
    return s


# ------------------------------------------------------------------------------------------------ selective scan
@pytest.mark.parametrize('B,L,Di,N', [(2, 37, 64, 8), (3, 150, 96, 16), (2, 129, 128, 32), (1, 70, 64, 64), (2, 64, 64, 32), (1, 5, 8, 4)])
@pytest.mark.parametrize('with_z', [True, False])
def test_selective_scan_fwd_bwd(ops, B, L, Di, N, with_z):
    g = torch.Generator().manual_seed(B * 1000 + L + N)
    # strided operands: x | z halves of one [B, L, 2Di] tensor; delta_r | B | C slices of one [B, L, R+2N] tensor
    xz = rnd(B, L, 2 * Di, g=g)
    R = 8
    xdbl = rnd(B, L, R + 2 * N, g=g)
    delta = rnd(B, L, Di, g=g, scale=0.5)
    A = -torch.exp(rnd(Di, N, g=g, scale=0.4))
    D, db = rnd(Di, g=g), rnd(Di, g=g, scale=0.2)
    start = make_start(B, L, g)
    dout = rnd(B, L, Di, g=g)

    def run(dev, fn):
        ts = [t.clone().to(dev).requires_grad_(True) for t in (xz, xdbl, delta, A, D, db)]
        xz_, xdbl_, delta_, A_, D_, db_ = ts
        u, z = xz_[..., :Di], (xz_[..., Di:] if with_z else None)
        Bm, Cm = xdbl_[..., R:R + N], xdbl_[..., R + N:]
        out, last = fn(u, delta_, A_, Bm, Cm, D_, z, db_, start.to(dev))
        (out * dout.to(dev)).sum().backward()
        return out, last, [t.grad for t in ts]

    o_ref, l_ref, g_ref = run('cpu', lambda u, d, A_, Bm, Cm, D_, z, db_, s: K.selective_scan_ref(u, d, A_, Bm, Cm, D_, z, db_, s, True))
    o_gpu, l_gpu, g_gpu = run('cuda', lambda u, d, A_, Bm, Cm, D_, z, db_, s: ops.selective_scan_tm(u, d, A_, Bm, Cm, D_, z, db_, s, True, True))
    close_fwd(o_gpu, o_ref, name='out')
    close_fwd(l_gpu, l_ref, name='last_state')
    for nm, a, b in zip(('dxz', 'dxdbl', 'ddelta', 'dA', 'dD', 'ddelta_bias'), g_gpu, g_ref):
        close(a, b, rtol=2e-4, atol_scale=5e-5, name=nm)


def _slow_decay_case(B, L, Di, N, seed):
    """Mamba's real initialisation range: A in [-1, -0.01] (log-uniform), delta = softplus(pre + bias) in [1e-3, 0.1] (log-uniform,
    mamba_simple's dt_min / dt_max) - a state survives hundreds of steps, so every carry across chunks, checkpoints and time
    segments contributes at full weight (with delta ~ 0.5, A ~ -1 a 64-step segment decays by e^-30 and the carry is invisible)."""
    g = torch.Generator().manual_seed(seed)
    u, z = rnd(B, L, Di, g=g), rnd(B, L, Di, g=g)
    Bm, Cm = rnd(B, L, N, g=g), rnd(B, L, N, g=g)
    A = -torch.exp(torch.rand(Di, N, generator=g) * math.log(100.0) + math.log(0.01))
    db = rnd(Di, g=g, scale=0.1)
    dt = torch.exp(torch.rand(B, L, Di, generator=g) * math.log(100.0) + math.log(1e-3))
    delta = torch.log(torch.expm1(dt)) - db                     # softplus(delta + bias) = dt
    D = rnd(Di, g=g)
    start = torch.zeros(B, L)
    start[:, 0] = 1
    if L > 700:
        start[0, 700] = 1                                       # one reset deep inside a row; the other rows run L steps unbroken
    w = rnd(B, L, Di, g=g)
    return (u, delta, A, Bm, Cm, D, z, db), start, w


def test_selective_scan_three_workgroup_forward_edition(ops):
    """Grids of 768 workgroups and more (B >= 96 rows at d_inner 512: configs[3]'s global batch on one GPU) run the TC = 16 forward
    (`sscan_fwd2_kernel<8, 4, 16, 0, 3>`: 162 VGPRs, 28 KB of LDS, three workgroups per CU) - same arithmetic, same 8-step checkpoints: output
    and every gradient (the backward replays from those checkpoints) against the oracle."""
    from oracle import kernels as K
    g = torch.Generator().manual_seed(5)
    B, L, Di, N = 96, 41, 512, 32
    r = lambda *s: torch.randn(*s, generator=g)
    u, delta, z, Bm, Cm = r(B, L, Di), r(B, L, Di) * 0.5, r(B, L, Di), r(B, L, N), r(B, L, N)
    A, D, db = -torch.exp(r(Di, N) * 0.3), r(Di), r(Di) * 0.1
    start = torch.zeros(B, L)
    start[:, 0] = 1
    start[::3, 17] = 1
    cpu = [t.clone().requires_grad_(True) for t in (u, delta, A, Bm, Cm, D, z, db)]
    ref, _ = K.selective_scan_ref(*cpu, start, True)
    w = r(B, L, Di)
    (ref * w).sum().backward()
    dev = [t.clone().cuda().requires_grad_(True) for t in (u, delta, A, Bm, Cm, D, z, db)]
    out = ops.selective_scan_tm(*dev, start.cuda(), True)
    (out * w.cuda()).sum().backward()
    close(out, ref, name='sscan fwd (TC 16)')
    for name, a, b in zip(('du', 'ddelta', 'dA', 'dB', 'dC', 'dD', 'dz', 'dbias'), dev, cpu):
        close(a.grad, b.grad, rtol=2e-4, name=name)


@pytest.mark.parametrize('B,L,Di,N,segs', [(2, 1043, 64, 32, 1), (2, 1043, 64, 32, 11), (3, 1043, 64, 32, 0), (1, 1043, 128, 16, 1),
                                            (1, 1043, 128, 16, 5), (2, 300, 64, 64, 1), (2, 300, 64, 8, 3)])
def test_selective_scan_slow_decay_vs_oracle(ops, monkeypatch, B, L, Di, N, segs):
    """Forward, last state and every gradient against the oracle at the BASELINE row length with slowly decaying states, with the
    one-pass kernels forced (segs = 1: the B = 64 bench path), the time-parallel kernels forced (segs > 1) and the library's
    own choice (segs = 0)."""
    ins, start, w = _slow_decay_case(B, L, Di, N, seed=L + N + segs)
    ref_in = [t.clone().requires_grad_(True) for t in ins]
    ref, ref_last = K.selective_scan_ref(*ref_in, start, True)
    (ref * w).sum().backward()
    monkeypatch.setattr(ops, 'SSCAN_TIME_SEGMENTS', segs)
    dev_in = [t.clone().cuda().requires_grad_(True) for t in ins]
    out, last = ops.selective_scan_tm(*dev_in, start.cuda(), True, return_last_state=True)
    (out * w.cuda()).sum().backward()
    # the carried state matters: the output with the state cut at every 32-step chunk would be far away from the reference
    assert ref_last.abs().max().item() > 1.0
    close_fwd(out, ref, name='out')
    close_fwd(last, ref_last, name='last_state')
    for a, b, nm in zip(dev_in, ref_in, ('du', 'ddelta', 'dA', 'dB', 'dC', 'dD', 'dz', 'dbias')):
        close(a.grad, b.grad, rtol=2e-4, atol_scale=5e-5, name=nm)


def test_selective_scan_slow_decay_carry_is_load_bearing(ops, monkeypatch):
    """The test data above does exercise the cross-segment carry: zeroing the state at the segment edges (extra resets at the
    11 segment starts) changes the output by far more than the parity tolerance, and the segmented kernels agree with the
    one-pass kernels to 1e-5 WITHOUT those resets."""
    ins, start, w = _slow_decay_case(2, 1043, 64, 32, seed=5)
    dev = [t.cuda() for t in ins]
    outs = {}
    for segs in (1, 11):
        monkeypatch.setattr(ops, 'SSCAN_TIME_SEGMENTS', segs)
        outs[segs] = ops.selective_scan_tm(*dev, start.cuda(), True)
    close(outs[11], outs[1].cpu(), rtol=1e-5, atol_scale=1e-6, name='segmented vs one pass')
    cut = start.clone()
    cut[:, ::96] = 1                                              # 11 segments of 96 steps at L = 1043
    monkeypatch.setattr(ops, 'SSCAN_TIME_SEGMENTS', 1)
    o_cut = ops.selective_scan_tm(*dev, cut.cuda(), True)
    rel = (o_cut - outs[1]).abs().max().item() / outs[1].abs().max().item()
    assert rel > 1e-2, f'cutting the carry changed the output by only {rel:.2e}'


def test_selective_scan_golden_reference_vectors(ops):
    """The reference's own selective_scan_ref outputs (tests/golden/selective_scan.npz), channel-major signature."""
    gold = load_golden('selective_scan.npz')
    for case in ('n16', 'n32', 'n64'):
        T = lambda k: torch.from_numpy(gold[f'{case}|{k}']).cuda()
        u, delta, z, Bm, Cm = [T(k).requires_grad_(True) for k in ('u', 'delta', 'z', 'Bm', 'Cm')]
        A, D, db = [T(k).requires_grad_(True) for k in ('A', 'D', 'delta_bias')]
        start = T('start')[:, None, :].expand(-1, u.shape[1], -1)
        out, last = ops.selective_scan_fn(u, delta, A, Bm, Cm, start, D, z, db, True, True)
        close_fwd(out, torch.from_numpy(gold[f'{case}|out']), name=f'{case} out')
        close_fwd(last, torch.from_numpy(gold[f'{case}|last_state']), name=f'{case} last')
        (out * T('dout')).sum().backward()
        for k, t in dict(u=u, delta=delta, z=z, Bm=Bm, Cm=Cm, A=A, D=D, delta_bias=db).items():
            close(t.grad, torch.from_numpy(gold[f'{case}|d{k}']), rtol=3e-4, atol_scale=1e-4, name=f'{case} d{k}')


def test_selective_scan_reset_isolation(ops):
    """Size-independent property at BASELINE config-2 width: outputs after a reset do not depend on inputs before it."""
    g = torch.Generator().manual_seed(3)
    B, L, Di, N = 2, 300, 512, 32
    mk = lambda *s: rnd(*s, g=g).cuda()
    u, delta, z, Bm, Cm = mk(B, L, Di), mk(B, L, Di) * 0.5, mk(B, L, Di), mk(B, L, N), mk(B, L, N)
    A = -torch.exp(mk(Di, N) * 0.3)
    start = torch.zeros(B, L, device='cuda')
    start[:, 0] = 1
    start[:, 170] = 1
    o1 = ops.selective_scan_tm(u, delta, A, Bm, Cm, None, z, None, start, True)
    u2 = u.clone()
    u2[:, :170] += 5.0
    o2 = ops.selective_scan_tm(u2, delta, A, Bm, Cm, None, z, None, start, True)
    assert torch.equal(o1[:, 170:], o2[:, 170:])
    assert not torch.equal(o1[:, :170], o2[:, :170])
    # bitwise reproducible (no atomics)
    assert torch.equal(o1, ops.selective_scan_tm(u, delta, A, Bm, Cm, None, z, None, start, True))


# ------------------------------------------------------------------------------------------------ conv1d
@pytest.mark.parametrize('B,L,Di,Kw', [(2, 50, 64, 4), (3, 130, 96, 16), (1, 70, 128, 3), (2, 65, 64, 8), (1, 200, 64, 20),
                                        (2, 64, 72, 8), (1, 300, 520, 16), (3, 129, 256, 2),
                                        (2, 192, 64, 16), (1, 1043, 128, 16), (2, 96, 64, 4), (2, 175, 64, 8)])       # the backward's 64 / 80 / 96-step chunks
def test_causal_conv1d_fwd_bwd(ops, B, L, Di, Kw):
    g = torch.Generator().manual_seed(L + Kw)
    xz = rnd(B, L, 2 * Di, g=g)
    w, b = rnd(Di, 1, Kw, g=g, scale=0.3), rnd(Di, g=g, scale=0.2)
    mask = (torch.rand(B, L, generator=g) > 0.2).float()
    dy = rnd(B, L, Di, g=g)

    def run(dev, fn):
        ts = [t.clone().to(dev).requires_grad_(True) for t in (xz, w, b)]
        y = fn(ts[0][..., :Di], ts[1], ts[2], mask.to(dev))
        (y * dy.to(dev)).sum().backward()
        return y, [t.grad for t in ts]

    y_ref, g_ref = run('cpu', lambda x, w_, b_, m: K.causal_conv1d_silu_ref(x, w_[:, 0], b_, m))
    y_gpu, g_gpu = run('cuda', lambda x, w_, b_, m: ops.causal_conv1d_fn(x, w_, b_, m, True))
    close_fwd(y_gpu, y_ref, name='y')
    for nm, a, b_ in zip(('dx', 'dw', 'db'), g_gpu, g_ref):
        close(a, b_, rtol=2e-4, atol_scale=5e-5, name=nm)


# ------------------------------------------------------------------------------------------------ add + norm
@pytest.mark.parametrize('M,C', [(37, 256), (100, 64), (9, 512), (5, 1024)])
@pytest.mark.parametrize('rms', [False, True])
@pytest.mark.parametrize('prenorm', [True, False])
def test_add_layernorm_fwd_bwd(ops, M, C, rms, prenorm):
    g = torch.Generator().manual_seed(M + C)
    x, r = rnd(2, M, C, g=g), rnd(2, M, C, g=g)
    w, b = 1 + rnd(C, g=g, scale=0.1), rnd(C, g=g, scale=0.1)
    dy, dr = rnd(2, M, C, g=g), rnd(2, M, C, g=g)

    def run(dev, fn):
        ts = [t.clone().to(dev).requires_grad_(True) for t in (x, r, w, b)]
        y, res = fn(ts[0], ts[1], ts[2], None if rms else ts[3])
        loss = (y * dy.to(dev)).sum()
        if prenorm:
            loss = loss + (res * dr.to(dev)).sum()
        loss.backward()
        return y, res, [t.grad for t in (ts[:3] if rms else ts)]

    y_ref, res_ref, g_ref = run('cpu', lambda x_, r_, w_, b_: K.add_layernorm_ref(x_, r_, w_, b_, 1e-8, rms))

    def gpu_fn(x_, r_, w_, b_):
        f = ops.rms_norm_fn if rms else ops.layer_norm_fn
        out = f(x_, w_, b_, residual=r_, eps=1e-8, prenorm=prenorm, residual_in_fp32=True)
        return out if prenorm else (out, x_ + r_)

    y_gpu, res_gpu, g_gpu = run('cuda', gpu_fn)
    close_fwd(y_gpu, y_ref, name='y')
    close_fwd(res_gpu, res_ref, name='res')
    for i, (a, b_) in enumerate(zip(g_gpu, g_ref)):
        close(a, b_, rtol=2e-4, atol_scale=5e-5, name=f'grad{i}')


@pytest.mark.parametrize('M,C,bias', [(4100, 256, True), (333, 512, True), (1000, 96, False)])
def test_add_layernorm_with_the_elu_epilogue_vs_torch(ops, M, C, bias):
    """act='elu' of the fused add + LayerNorm: elu(LN(x + residual)) and its gradients (elu' re-formed from the recomputed pre-activation)."""
    g = torch.Generator().manual_seed(M + C)
    x, r, dy = rnd(M, C, g=g), rnd(M, C, g=g), rnd(M, C, g=g)
    w, b = 1 + 0.2 * rnd(C, g=g), (0.3 * rnd(C, g=g) if bias else None)

    def run(dev, fn):
        ts = [t.clone().to(dev).requires_grad_(True) for t in ((x, r, w, b) if bias else (x, r, w))]
        y = fn(*ts) if bias else fn(ts[0], ts[1], ts[2], None)
        y.backward(dy.to(dev))
        return y, [t.grad for t in ts]

    y_ref, g_ref = run('cpu', lambda x_, r_, w_, b_: torch.nn.functional.elu(torch.nn.functional.layer_norm(x_ + r_, (C,), w_, b_, 1e-5)))
    y_gpu, g_gpu = run('cuda', lambda x_, r_, w_, b_: ops.layer_norm_fn(x_, w_, b_, residual=r_, eps=1e-5, act='elu'))
    close_fwd(y_gpu, y_ref, name='y')
    for nm, a, b_ in zip(('dx', 'dres', 'dw', 'db'), g_gpu, g_ref):
        close(a, b_, rtol=2e-4, atol_scale=5e-5, name=nm)


@pytest.mark.parametrize('second_in_place', [True, False])
def test_row_buffer_producers_write_in_place_vs_cat(ops, second_in_place):
    """Two `linear_act` producers (one with ELU: its saved output is a VIEW of the buffer) write the column blocks of an `ops.RowBuffer`;
    `ops.cat_into` hands the buffer on without a copy, tagged with the shared magnitude handle.  Same values and gradients as producing
    the two outputs separately and `torch.cat`; a piece that did not land in place is copied in (and the result carries no handle)."""
    g = torch.Generator().manual_seed(21)
    B, L = 5, 1000
    x1, x2 = rnd(B, L, 64, g=g).cuda(), rnd(B, L, 128, g=g).cuda()
    w1, b1, w2, b2 = rnd(256, 64, g=g).cuda() / 8, rnd(256, g=g).cuda(), rnd(128, 128, g=g).cuda() / 11, rnd(128, g=g).cuda()
    gout = rnd(B, L, 384, g=g).cuda()

    def leafs():
        return [t.clone().requires_grad_(True) for t in (x1, w1, b1, x2, w2, b2)]

    a = leafs()
    rb = ops.RowBuffer((B, L), 384, 'cuda')
    y1 = ops.linear_act(a[0], a[1], a[2], 'elu', dest=rb.block(0, 256))
    y2 = ops.linear_act(a[3], a[4], a[5], None, dest=rb.block(256, 128) if second_in_place else None)
    assert rb.block(0, 256).holds(y1) and rb.block(256, 128).holds(y2) == second_in_place
    cat = ops.cat_into(rb, [(y1, 0), (y2, 256)])
    assert cat.shape == (B, L, 384) and cat.data_ptr() == rb.buf.data_ptr()
    (cat * gout).sum().backward()
    r = leafs()
    ref = torch.cat((ops.linear_act(r[0], r[1], r[2], 'elu'), ops.linear_act(r[3], r[4], r[5], None)), dim=-1)
    (ref * gout).sum().backward()
    assert torch.equal(cat, ref)
    for p, q in zip(a, r):
        assert torch.equal(p.grad, q.grad)
    h = ops.amax_of(cat)
    if second_in_place:
        assert h is not None and ops.amax_value(h) == float(ref.detach().abs().max())
    else:
        assert h is None


def test_place_blocks_vs_block_diag_cat_pad(ops):
    """`ops.place_blocks` (one launch) against torch.block_diag / cat / pad, values and gradients (the operands of the merged encoder GEMM,
    policy_value_models/_inputs.py)."""
    g = torch.Generator().manual_seed(11)
    ws = [rnd(128, 17, g=g), rnd(128, 17, g=g), rnd(128, 6, g=g), rnd(128, 1, g=g)]
    bs = [rnd(128, g=g) for _ in ws]
    batch = rnd(5, 301, 60, g=g).cuda()
    xs_src = [batch[..., 0:17], batch[..., 20:37], batch[..., 40:46], batch[..., 50:51]]       # column blocks of one packed batch array
    ks, ns = [17, 17, 6, 1], [128] * 4
    kp = 44
    c0 = [0, 17, 34, 40]
    r0 = [0, 128, 256, 384]

    def leafs(ts):
        return [t.clone().cuda().requires_grad_(True) for t in ts]

    w1, b1, x1 = leafs(ws), leafs(bs), [t.clone().requires_grad_(True) for t in xs_src]
    W = ops.place_blocks(512, kp, list(zip(r0, c0)), *w1)
    Bc = ops.place_blocks(1, 512, [(0, r) for r in r0], *b1).view(-1)
    X = ops.place_blocks(5 * 301, kp, [(0, c) for c in c0], *x1).view(5, 301, kp)
    w2, b2, x2 = leafs(ws), leafs(bs), [t.clone().requires_grad_(True) for t in xs_src]
    Wr = torch.nn.functional.pad(torch.block_diag(*w2), (0, kp - 41))
    Br = torch.cat(b2)
    Xr = torch.cat(x2 + [torch.zeros(5, 301, kp - 41, device='cuda')], dim=-1)
    assert torch.equal(W, Wr) and torch.equal(Bc, Br) and torch.equal(X, Xr)
    gw, gb, gx = rnd(512, kp, g=g).cuda(), rnd(512, g=g).cuda(), rnd(5, 301, kp, g=g).cuda()
    ((W * gw).sum() + (Bc * gb).sum() + (X * gx).sum()).backward()
    ((Wr * gw).sum() + (Br * gb).sum() + (Xr * gx).sum()).backward()
    for a, b in zip(w1 + b1 + x1, w2 + b2 + x2):
        assert torch.equal(a.grad, b.grad)


# ------------------------------------------------------------------------------------------------ linear recurrences
# (64, 40, 256): 64 channels per wave; (16, 45, 512): 32 x 2 time segments; the small batches: 16 channels x 4 segments per wave
@pytest.mark.parametrize('B,L,C', [(2, 33, 64), (3, 500, 96), (1, 2003, 256), (2, 7, 32), (64, 40, 256), (16, 45, 512)])
@pytest.mark.parametrize('fuse', [True, False])
def test_gilr_scan_fwd_bwd(ops, B, L, C, fuse):
    g = torch.Generator().manual_seed(L + C)
    v, f = rnd(B, L, C, g=g), rnd(B, L, C, g=g)
    if not fuse:
        f = torch.sigmoid(f)
    start = make_start(B, L, g)
    h0 = rnd(B, C, g=g)
    dh = rnd(B, L, C, g=g)

    def run(dev, fn):
        ts = [t.clone().to(dev).requires_grad_(True) for t in (v, f)]
        h = fn(ts[0], ts[1], start.to(dev), h0.to(dev))
        (h * dh.to(dev)).sum().backward()
        return h, [t.grad for t in ts]

    h_ref, g_ref = run('cpu', lambda v_, f_, s, h: K.linrec_real_ref(v_, f_, s, h, fuse)[0])
    h_gpu, g_gpu = run('cuda', lambda v_, f_, s, h: ops.gilr_scan(v_, f_, s, h, fuse))
    close_fwd(h_gpu, h_ref, name='h')
    close(g_gpu[0], g_ref[0], rtol=2e-4, atol_scale=5e-5, name='dv')
    close(g_gpu[1], g_ref[1], rtol=2e-4, atol_scale=5e-5, name='df')


@pytest.mark.parametrize('B,L,C', [(2, 33, 64), (3, 500, 96), (1, 2003, 256), (64, 40, 256), (16, 45, 512)])
def test_lru_scan_fwd_bwd(ops, B, L, C):
    g = torch.Generator().manual_seed(L + C + 1)
    vr, vi = rnd(B, L, C, g=g), rnd(B, L, C, g=g)
    mag, th = 0.9 + 0.099 * torch.rand(C, generator=g), 6.28 * torch.rand(C, generator=g)
    lr, li = mag * torch.cos(th), mag * torch.sin(th)
    gamma = torch.sqrt(1 - mag ** 2)
    start = make_start(B, L, g)
    dhr, dhi = rnd(B, L, C, g=g), rnd(B, L, C, g=g)

    def run(dev, fn):
        ts = [t.clone().to(dev).requires_grad_(True) for t in (vr, vi, lr, li, gamma)]
        hr, hi = fn(*ts, start.to(dev))
        ((hr * dhr.to(dev)).sum() + (hi * dhi.to(dev)).sum()).backward()
        return hr, hi, [t.grad for t in ts]

    r_ref = run('cpu', lambda a, b, c, d, e, s: K.linrec_complex_ref(a, b, c, d, s, gamma=e))
    r_gpu = run('cuda', lambda a, b, c, d, e, s: ops.complex_scan(a, b, c, d, e, s))
    close_fwd(r_gpu[0], r_ref[0], name='hr')
    close_fwd(r_gpu[1], r_ref[1], name='hi')
    for nm, a, b in zip(('dvr', 'dvi', 'dlam_re', 'dlam_im', 'dgamma'), r_gpu[2], r_ref[2]):
        close(a, b, rtol=3e-4, atol_scale=1e-4, name=nm)


@pytest.mark.parametrize('B,L,C', [(2, 33, 64), (16, 301, 256), (3, 50, 96)])
@pytest.mark.parametrize('layout', ['shared', 'dense'])
def test_linrec_members_in_place_equal_the_dense_forms(ops, B, L, C, layout):
    """`gilr_scan_members` / `complex_scan_members` read the members of u [E, B, T, C] through their row stride - u being the [E, B, T, C]
    view of a shared-input EnsembleLinear's [M, E C] output (reference gilr.py:60-62, lru.py:112-120) - and hand back ONE gradient tensor in
    u's layout: same bits as the dense per-member calls, against the oracle like them."""
    g = torch.Generator().manual_seed(B * L + C)
    start = make_start(B, L, g).cuda()

    def members(E):
        y2 = rnd(B * L, E * C, g=g).cuda()
        if layout == 'shared':
            return y2.view(B * L, E, C).transpose(0, 1).reshape(E, B, L, C)        # strides (C, L E C, E C, 1): a view
        return y2.view(B * L, E, C).transpose(0, 1).contiguous().view(E, B, L, C)

    # ---- real (gilr)
    u = members(2).requires_grad_(True)
    h0, dh = rnd(B, C, g=g).cuda(), rnd(B, L, C, g=g).cuda()
    h = ops.gilr_scan_members(u, start, h0, True)
    (du,) = torch.autograd.grad(h, u, dh)
    assert du.stride() == u.stride()
    vd, fd = u.detach()[0].contiguous().requires_grad_(True), u.detach()[1].contiguous().requires_grad_(True)
    hd = ops.gilr_scan(vd, fd, start, h0, True)
    dv, df = torch.autograd.grad(hd, (vd, fd), dh)
    assert torch.equal(h, hd) and torch.equal(du[0], dv) and torch.equal(du[1], df)
    assert ops.amax_of(h) is not None or h.numel() < (1 << 20)
    if h.numel() >= (1 << 20):
        assert ops.amax_value(ops.amax_of(h)) == float(h.detach().abs().max()) and ops.amax_value(ops.amax_of(du)) == float(du.abs().max())
    vs, fs = u.detach()[0], u.detach()[1]                                          # the strided members given separately: read in place as well
    assert torch.equal(ops.gilr_scan(vs, fs, start, h0, True), hd)
    h_ref = K.linrec_real_ref(vd.detach().cpu(), fd.detach().cpu(), start.cpu(), h0.cpu(), True)[0]
    close_fwd(h.cpu(), h_ref, name='h')

    # ---- complex (lru), E = 3 with the pass-through member and the (Re | Im) combination
    mag, th = 0.9 + 0.099 * torch.rand(C, generator=g), 6.28 * torch.rand(C, generator=g)
    lr, li, gm = (mag * torch.cos(th)).cuda(), (mag * torch.sin(th)).cuda(), torch.sqrt(1 - mag ** 2).cuda()
    u = members(3).requires_grad_(True)
    dh2, dout = rnd(2, B, L, C, g=g).cuda(), rnd(B, L, C, g=g).cuda()
    lam3 = torch.stack((lr, li, gm)).requires_grad_(True)
    h2, u2 = ops.complex_scan_members(u, lam3, start)
    out = ops.SubAddMembers.apply(h2, u2)
    grads = torch.autograd.grad((h2 * dh2).sum() + (out * dout).sum(), [u, lam3])
    assert grads[0].stride() == u.stride()
    ud = [u.detach()[e].contiguous().requires_grad_(True) for e in range(3)]
    pars_d = [t.clone().requires_grad_(True) for t in (lr, li, gm)]
    hr, hi = ops.complex_scan(ud[0], ud[1], *pars_d, start)
    out_d = hr - hi + ud[2]
    grads_d = torch.autograd.grad((hr * dh2[0]).sum() + (hi * dh2[1]).sum() + (out_d * dout).sum(), ud + pars_d)
    assert torch.equal(h2[0], hr) and torch.equal(h2[1], hi) and torch.equal(out, out_d)
    for e in range(3):
        close(grads[0][e], grads_d[e].cpu(), rtol=1e-6, atol_scale=1e-6, name=f'du{e}')      # (dh2 + dout) is summed in another order
    for i, b in enumerate(grads_d[3:]):
        close(grads[1][i], b.cpu(), rtol=1e-5, atol_scale=1e-5, name='dparam')


def test_lru_params_one_launch_vs_torch(ops):
    """lam3 = (lam_re | lam_im | gamma) from params_log (reference lru.py:104-110) and its gradient against the same formula in torch."""
    g = torch.Generator().manual_seed(5)
    for C in (256, 96, 1000):
        p = (torch.randn(3, C, generator=g) * 0.5 - 1.0)
        d = torch.randn(3, C, generator=g)
        pr = p.clone().requires_grad_(True)
        nu, theta, gamma = torch.exp(pr)
        mag = torch.exp(-nu)
        ref = torch.stack((mag * torch.cos(theta), mag * torch.sin(theta), gamma))
        (gref,) = torch.autograd.grad(ref, pr, d)
        pg = p.cuda().requires_grad_(True)
        out = ops.lru_params(pg)
        (gg,) = torch.autograd.grad(out, pg, d.cuda())
        close(out, ref, rtol=2e-6, atol_scale=1e-6, name='lam3')
        close(gg, gref, rtol=5e-6, atol_scale=2e-6, name='dparams_log')


# ------------------------------------------------------------------------------------------------ GRU
@pytest.mark.parametrize('B,L,H', [(3, 20, 64), (18, 40, 256), (2, 130, 32), (5, 9, 80), (2, 6, 384), (68, 6, 256)])
def test_gru_seq_fwd_bwd_vs_aten(ops, B, L, H):
    """Against torch.nn.GRU on CPU (the reference's GRU layer is exactly that module, rnn_base.py:59)."""
    g = torch.Generator().manual_seed(H + L)
    gru = torch.nn.GRU(H, H, batch_first=True)
    with torch.no_grad():
        for p in gru.parameters():
            p.copy_(rnd(*p.shape, g=g, scale=1 / math.sqrt(H)))
    x = rnd(B, L, H, g=g)
    dy = rnd(B, L, H, g=g)
    xr = x.clone().requires_grad_(True)
    y_ref, _ = gru(xr)
    (y_ref * dy).sum().backward()
    ps = {n: p.detach().clone().cuda().requires_grad_(True) for n, p in gru.named_parameters()}
    xg = x.clone().cuda().requires_grad_(True)
    gi = torch.nn.functional.linear(xg, ps['weight_ih_l0'], ps['bias_ih_l0'])
    y = ops.gru_seq(gi, ps['weight_hh_l0'], ps['bias_hh_l0'])
    (y * dy.cuda()).sum().backward()
    close_fwd(y, y_ref, name='h_all')
    close(xg.grad, xr.grad, rtol=2e-4, atol_scale=5e-5, name='dx')
    for n, p in gru.named_parameters():
        close(ps[n].grad, p.grad, rtol=3e-4, atol_scale=1e-4, name=n)
    # the oracle's explicit-formula GRU agrees with ATen too
    y_or = K.gru_seq_ref(torch.nn.functional.linear(x, gru.weight_ih_l0, gru.bias_ih_l0), gru.weight_hh_l0, gru.bias_hh_l0)
    close_fwd(y, y_or.detach(), name='h_all vs oracle')


# ------------------------------------------------------------------------------------------------ SAC arithmetic
def test_tanh_gaussian_fwd_bwd(ops):
    g = torch.Generator().manual_seed(1)
    out2 = rnd(5, 33, 12, g=g, scale=2.0)
    out2[0, 0, :6] = 5.0       # clamped log-std (no gradient through the clamp)
    out2[0, 1, :6] = -30.0
    noise = rnd(5, 33, 6, g=g)
    ds, dl = rnd(5, 33, 6, g=g), rnd(5, 33, 1, g=g)

    def run(dev, fn):
        o = out2.clone().to(dev).requires_grad_(True)
        mean, samp, logp = fn(o, noise.to(dev))
        ((samp * ds.to(dev)).sum() + (logp * dl.to(dev)).sum()).backward()
        return mean, samp, logp, o.grad

    ref = run('cpu', lambda o, n: K.tanh_gaussian_ref(o[..., 6:], o[..., :6], n))
    got = run('cuda', ops.tanh_gaussian)
    for nm, a, b in zip(('mean', 'sample', 'logp', 'dout2'), got, ref):
        close(a, b, rtol=2e-4, atol_scale=5e-5, name=nm)


def test_sac_target_and_guard(ops):
    g = torch.Generator().manual_seed(2)
    E, R, L = 8, 6, 50
    q = rnd(E, R, L, 1, g=g, scale=3.0)
    logp, reward = rnd(R, L, 1, g=g), rnd(R, L, 1, g=g)
    done = (torch.rand(R, L, 1, generator=g) < 0.05).float()
    mask = (torch.rand(R, L, 1, generator=g) < 0.8).float()
    log_alpha = torch.tensor([-0.3])
    guard = torch.tensor([1e6, -1e6, 0.0, 1 - 1e-3]).cuda()
    stats = torch.zeros(2).cuda()
    gmin = gmax = None
    for it in range(3):
        idx = torch.randperm(E, generator=g)[:2]
        v = q[idx].min(dim=0).values - log_alpha.exp() * logp
        if gmin is None:
            gmin, gmax = v.min().item(), v.max().item()
        ref = K.sac_target_ref(q[idx], logp, reward, done, log_alpha.exp(), 0.99, gmin, gmax)
        got = ops.sac_target(q.cuda(), idx.int().cuda(), logp.cuda(), log_alpha.cuda(), reward.cuda(), done.cuda(), mask.cuda(),
                             0.99, guard, stats)
        close(got, ref, name=f'target[{it}]')
        ym = ref * mask                                   # q_value_guard.py:29-38
        gmin, gmax = min(gmin, ym.min().item()), max(gmax, ym.max().item())
        gmin = (1 - 1e-3) * gmin + 1e-3 * ym.min().item()
        gmax = (1 - 1e-3) * gmax + 1e-3 * ym.max().item()
        gd = guard.cpu()
        assert gd[0].item() == pytest.approx(gmin, rel=1e-5, abs=1e-5) and gd[1].item() == pytest.approx(gmax, rel=1e-5, abs=1e-5)
        st = stats.cpu()
        assert st[0].item() == pytest.approx(ref.abs().max().item(), rel=1e-5)
        assert st[1].item() == mask.sum().item()
        q = q + 0.1 * rnd(E, R, L, 1, g=g)


def test_flat_optimizer_tail(ops):
    g = torch.Generator().manual_seed(4)
    n = 10007
    p0, gr = rnd(n, g=g), rnd(n, g=g)
    tgt = rnd(n, g=g)
    t2 = tgt.clone().cuda()
    ops.soft_update_(t2, p0.cuda(), 0.995)
    close(t2, K.soft_update_ref(tgt, p0, 0.995), name='soft_update')
    close(ops.sumsq(p0.cuda()), (p0 ** 2).sum().reshape(1), name='sumsq')
    # AdamW with two learning-rate segments against torch.optim.AdamW
    cut = 4001
    a, b = p0[:cut].clone().requires_grad_(True), p0[cut:].clone().requires_grad_(True)
    opt = torch.optim.AdamW([{'params': [a], 'lr': 1e-2, 'weight_decay': 0.0}, {'params': [b], 'lr': 3e-3, 'weight_decay': 0.01}])
    pg, m, v = p0.clone().cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    seg_end = torch.tensor([cut, n], dtype=torch.int64).cuda()
    seg_lr, seg_wd = torch.tensor([1e-2, 3e-3]).cuda(), torch.tensor([0.0, 0.01]).cuda()
    for step in range(1, 4):
        gstep = gr * step
        a.grad, b.grad = gstep[:cut].clone(), gstep[cut:].clone()
        opt.step()
        ops.adamw_flat_(pg, gstep.cuda(), m, v, seg_end, seg_lr, seg_wd, step)
    close(pg, torch.cat((a, b)).detach(), rtol=1e-5, atol_scale=1e-6, name='adamw')


# ------------------------------------------------------------------------------------------------ cgpt attention (bf16)
@pytest.mark.parametrize('H,hd,lens', [(2, 32, [5]), (8, 32, [1, 130, 37, 64]), (4, 64, [200, 33]), (8, 32, [1027])])
def test_attn_varlen_alibi_fwd_bwd(ops, H, hd, lens):
    """bf16 MFMA attention vs the fp32 oracle on bf16-rounded inputs (north_star: 1e-2 for the bf16 path).
    The oracle restates published flash-attn semantics (causal + ALiBi, bottom-right aligned = same-length q/k): parity unpinned."""
    g = torch.Generator().manual_seed(sum(lens) + H)
    T = sum(lens)
    qkv = (torch.randn(T, 3, H, hd, generator=g) * 0.8).to(torch.bfloat16)
    dout = torch.randn(T, H, hd, generator=g).to(torch.bfloat16)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
    slopes = K.alibi_slopes(H)
    ref_in = qkv.float().requires_grad_(True)
    ref = K.attention_alibi_varlen_ref(ref_in[:, 0], ref_in[:, 1], ref_in[:, 2], cu, slopes, p_bf16=True)
    (ref * dout.float()).sum().backward()
    x = qkv.cuda().requires_grad_(True)
    out = ops.attn_varlen(x, cu.cuda(), max(lens), slopes.cuda())
    (out.float() * dout.cuda().float()).sum().backward()
    close(out.float(), ref, rtol=5e-3, atol_scale=5e-3, name='out')            # 1e-2 in all (north_star's bf16 bar)
    close(x.grad.float(), ref_in.grad, rtol=1e-2, atol_scale=1e-2, name='dqkv')


def test_attn_ragged_batch_work_list(ops):
    """70 sequences (more than one work-list chunk) of lengths 1 .. 300, several exactly on / next to the 128-token block edges:
    every (sequence, block) item of the device-built work list is run once, and the result is the oracle's."""
    H, hd = 4, 32
    g = torch.Generator().manual_seed(11)
    lens = [1, 128, 129, 127, 256, 257, 300, 2, 64, 255] + [int(x) for x in torch.randint(1, 301, (60,), generator=g)]
    T = sum(lens)
    qkv = (torch.randn(T, 3, H, hd, generator=g) * 0.8).to(torch.bfloat16)
    dout = torch.randn(T, H, hd, generator=g).to(torch.bfloat16)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
    slopes = K.alibi_slopes(H)
    ref_in = qkv.float().requires_grad_(True)
    ref = K.attention_alibi_varlen_ref(ref_in[:, 0], ref_in[:, 1], ref_in[:, 2], cu, slopes, p_bf16=True)
    (ref * dout.float()).sum().backward()
    x = qkv.cuda().requires_grad_(True)
    out = ops.attn_varlen(x, cu.cuda(), max(lens), slopes.cuda())
    (out.float() * dout.cuda().float()).sum().backward()
    close(out.float(), ref, rtol=5e-3, atol_scale=5e-3, name='out')            # 1e-2 in all (north_star's bf16 bar)
    close(x.grad.float(), ref_in.grad, rtol=1e-2, atol_scale=1e-2, name='dqkv')


def test_attn_forward_without_work_list_is_bitwise_the_same(ops):
    """workspace == NULL: the kernels fall back to (sequence, block) order - scheduling only, the same bits."""
    from offpolicy_rnn.hip._lib import lib
    from offpolicy_rnn.hip.ops import _p, _stream, check
    H, hd, lens = 8, 32, [1, 130, 37, 300, 64]
    g = torch.Generator().manual_seed(3)
    T, S = sum(lens), len(lens)
    qkv = (torch.randn(T, 3, H, hd, generator=g) * 0.8).to(torch.bfloat16).cuda()
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32).cuda()
    slopes = K.alibi_slopes(H).cuda()
    with torch.no_grad():
        a = ops.attn_varlen(qkv, cu, max(lens), slopes)
    out = torch.empty(T, H, hd, dtype=torch.bfloat16, device='cuda')
    lse = torch.empty(H, T, dtype=torch.float32, device='cuda')
    check(lib().resel_attn_varlen_fwd(_p(qkv), _p(cu), _p(slopes), _p(out), _p(lse), None, T, S, H, hd, max(lens), hd ** -0.5,
                                      0.0, 0, 0, _stream()), 'attn_varlen_fwd')
    torch.cuda.synchronize()
    assert torch.equal(a.view(torch.int16), out.view(torch.int16))


def test_attn_more_heads_than_xcds_and_grid_bound_above_the_longest_sequence(ops):
    """H = 12 (two head groups per XCD slot of the work-item mapping) and max_seqlen well above the longest sequence (work-list levels
    beyond every sequence are simply empty)."""
    H, hd, lens = 12, 32, [3, 200, 129, 64, 1]
    g = torch.Generator().manual_seed(21)
    T = sum(lens)
    qkv = (torch.randn(T, 3, H, hd, generator=g) * 0.8).to(torch.bfloat16)
    dout = torch.randn(T, H, hd, generator=g).to(torch.bfloat16)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
    slopes = K.alibi_slopes(H)
    ref_in = qkv.float().requires_grad_(True)
    ref = K.attention_alibi_varlen_ref(ref_in[:, 0], ref_in[:, 1], ref_in[:, 2], cu, slopes, p_bf16=True)
    (ref * dout.float()).sum().backward()
    x = qkv.cuda().requires_grad_(True)
    out = ops.attn_varlen(x, cu.cuda(), 1000, slopes.cuda())
    (out.float() * dout.cuda().float()).sum().backward()
    close(out.float(), ref, rtol=5e-3, atol_scale=5e-3, name='out')            # 1e-2 in all (north_star's bf16 bar)
    close(x.grad.float(), ref_in.grad, rtol=1e-2, atol_scale=1e-2, name='dqkv')


def test_attn_more_sequences_than_the_work_list_can_index(ops):
    """S >= 65 536 sequences: no work list is built (16-bit sequence field), the kernels run in (sequence, block) order.  With one
    token per sequence the softmax is over a single key: out == v and dv == dout exactly; dq, dk = 0 up to the fp32 summation order of
    dP - delta (two sums of the same 32 products)."""
    S, H, hd = 70000, 2, 32
    g = torch.Generator().manual_seed(2)
    qkv = torch.randn(S, 3, H, hd, generator=g).to(torch.bfloat16).cuda().requires_grad_(True)
    cu = torch.arange(S + 1, dtype=torch.int32, device='cuda')
    out = ops.attn_varlen(qkv, cu, 1, None)
    assert torch.equal(out, qkv.detach()[:, 2])
    dout = torch.randn(S, H, hd, generator=g).to(torch.bfloat16).cuda()
    out.backward(dout)
    assert torch.equal(qkv.grad[:, 2], dout) and qkv.grad[:, :2].float().abs().max() < 1e-4


def test_attn_layout_with_integer_data(ops):
    """Exact small-integer operands (products and sums exact in bf16/fp32) catch any fragment-layout transposition."""
    H, hd, L = 2, 32, 70
    g = torch.Generator().manual_seed(0)
    q = torch.zeros(L, H, hd)
    k = torch.zeros(L, H, hd)
    v = torch.randint(-3, 4, (L, H, hd), generator=g).float()           # asymmetric V
    qkv = torch.stack((q, k, v), dim=1).to(torch.bfloat16)
    cu = torch.tensor([0, L], dtype=torch.int32)
    out = ops.attn_varlen(qkv.cuda(), cu.cuda(), L, None).float().cpu()   # zero scores, no alibi: out[i] = mean_{j<=i} v[j]
    ref = torch.cumsum(v, dim=0) / torch.arange(1, L + 1).view(L, 1, 1)
    assert (out - ref).abs().max() < 0.02


# ------------------------------------------------------------------------------------------ fused Mamba mixer node
@pytest.mark.parametrize('B,L,Dm,N,Kw', [(2, 70, 32, 32, 16), (3, 129, 64, 16, 4)])
def test_mamba_inner_fused_vs_oracle_chain(ops, B, L, Dm, N, Kw):
    """One autograd node for the whole mixer (reference MambaInnerFn, selective_scan_interface_new.py:169) against the
    chain of CPU oracle ops under torch autograd: output and every parameter / input gradient."""
    import oracle_backend as ob
    g = torch.Generator().manual_seed(11)
    Di, R = 2 * Dm, max(Dm // 16, 4)
    x = rnd(B, L, Dm, g=g)
    ps = dict(in_w=rnd(2 * Di, Dm, g=g, scale=Dm ** -0.5), conv_w=rnd(Di, 1, Kw, g=g, scale=0.3), conv_b=rnd(Di, g=g, scale=0.1),
              xproj_w=rnd(R + 2 * N, Di, g=g, scale=Di ** -0.5), dt_w=rnd(Di, R, g=g, scale=R ** -0.5), dt_b=rnd(Di, g=g, scale=0.5) - 2,
              A_log=torch.log(torch.arange(1, N + 1, dtype=torch.float32)).repeat(Di, 1), D=torch.ones(Di) + rnd(Di, g=g, scale=0.1),
              out_w=rnd(Dm, Di, g=g, scale=Di ** -0.5))
    start = make_start(B, L, g)
    mask = (torch.rand(B, L, 1, generator=g) > 0.1).float()
    w = rnd(B, L, Dm, g=g)

    def run(fn, dev):
        xs = x.to(dev).detach().clone().requires_grad_(True)
        pp = {k: v.to(dev).detach().clone().requires_grad_(True) for k, v in ps.items()}
        out = fn(xs, pp['in_w'], pp['conv_w'], pp['conv_b'], pp['xproj_w'], pp['dt_w'], pp['dt_b'], pp['A_log'], pp['D'], pp['out_w'],
                 mask.to(dev), start.to(dev))
        (out * w.to(dev)).sum().backward()
        return out, xs.grad, {k: v.grad for k, v in pp.items()}

    ref_out, ref_dx, ref_g = run(ob.mamba_inner_fn, 'cpu')
    out, dx, gr = run(ops.mamba_inner_fn, 'cuda')
    close_fwd(out, ref_out, name='out')
    close(dx, ref_dx, name='dx')
    for k in ps:
        close(gr[k], ref_g[k], rtol=2e-4, atol_scale=5e-5, name='d' + k)


# ------------------------------------------------------------------------------------------ bias + activation tail
@pytest.mark.parametrize('rows,C,nseg', [(300, 64, 1), (8 * 129, 256, 8), (1000, 2048, 1), (6 * 37, 12, 6), (7, 8, 1), (2 * 513, 260, 2)])
@pytest.mark.parametrize('act', ['elu', None])
def test_bias_act_fwd_bwd(ops, rows, C, nseg, act):
    """act(y + bias[segment]) in place and its backward from the output (fc / efc-E tail, reference rnn_base.py:462-474)."""
    g = torch.Generator().manual_seed(5)
    y, bias, go = rnd(rows, C, g=g, scale=2.0), rnd(nseg, C, g=g), rnd(rows, C, g=g)
    yr = y.clone().requires_grad_(True)
    br = bias.clone().requires_grad_(True)
    pre = yr.view(nseg, -1, C) + br.view(nseg, 1, C)
    ref = (torch.nn.functional.elu(pre) if act == 'elu' else pre).reshape(rows, C)
    ref.backward(go)
    a = ops.bias_act_(y.cuda(), bias.cuda(), rows // nseg, act)
    close(a, ref, name='act(y + b)')
    gy, db = ops.bias_act_bwd(go.cuda(), a, rows // nseg, act, True)
    close(gy, yr.grad, name='gy')
    close(db, br.grad, rtol=2e-4, atol_scale=5e-5, name='dbias')


@pytest.mark.parametrize('R,T,n_in,H', [(3, 50, 24, 32), (3, 1500, 128, 128)])
def test_ensemble_mlp_fused_tail_vs_torch(ops, R, T, n_in, H):
    """shared-input layer -> per-member layer -> per-member head, ELU fused into the first two (efc-8 critic head,
    reference contextual_sac_value.py via rnn_base.py:462-474) against plain torch autograd on the CPU.  The second size
    (4500 tokens, 128-wide) is past the thresholds of `ops.gemm_f32_ok`: every contraction of the three nodes - forward with the
    bias / ELU epilogue, input gradients, weight gradients - then runs in `resel_gemm_f32` instead of the library."""
    from offpolicy_rnn.models.ensemble_linear_model import EnsembleLinear
    torch.manual_seed(3)
    E = 4
    flops0 = ops.GEMM_FLOPS[0]
    l1, l2, l3 = EnsembleLinear(n_in, H, E), EnsembleLinear(H, H, E, desire_ndim=4), EnsembleLinear(H, 1, E, desire_ndim=4)
    for l in (l1, l2, l3):
        torch.nn.init.normal_(l.bias, std=0.3)
    x = torch.randn(R, T, n_in)
    w = torch.randn(E, R, T, 1)

    def ref():
        xs = x.clone().requires_grad_(True)
        h = xs
        for i, l in enumerate((l1, l2, l3)):
            h = torch.einsum('...ti,eio->e...to', h, l.weight) if i == 0 else torch.einsum('e...ti,eio->e...to', h, l.weight)
            h = h + l.bias.view(E, 1, 1, -1)
            if i < 2:
                h = torch.nn.functional.elu(h)
        (h * w).sum().backward()
        out = [h.detach(), xs.grad] + [p.grad.clone() for l in (l1, l2, l3) for p in (l.weight, l.bias)]
        for l in (l1, l2, l3):
            l.zero_grad()
        return out

    from offpolicy_rnn.models.ensemble_linear_model import ensemble_head, head_fusable
    expect = ref()
    for l in (l1, l2, l3):
        l.cuda()
    for fused_head in (False, True):                 # layer-by-layer tails, then the one-node head (hidden tail + width-1 output)
        xs = x.cuda().requires_grad_(True)
        h1 = l1(xs, act='elu')
        if fused_head:
            assert head_fusable(l2, torch.nn.ELU(), l3, torch.nn.Identity(), h1)
            h = ensemble_head(l2, l3, h1)
        else:
            h = l3(l2(h1, act='elu'))
        (h * w.cuda()).sum().backward()
        got = [h, xs.grad] + [p.grad for l in (l1, l2, l3) for p in (l.weight, l.bias)]
        for i, (a, b) in enumerate(zip(got, expect)):
            close(a, b, rtol=2e-4, atol_scale=5e-5, name=f'fused_head={fused_head} tensor {i}')
        for l in (l1, l2, l3):
            l.zero_grad()
    assert (ops.GEMM_FLOPS[0] > flops0) == (R * T >= ops.GEMM_F32_MIN_ROWS), 'hand-written GEMM routing'


@pytest.mark.parametrize('R,T,n_in,H1,H2,E', [(3, 1500, 128, 128, 128, 4), (4, 1043, 384, 256, 256, 8), (5, 1024, 96, 128, 160, 2)])
@pytest.mark.parametrize('part', [False, True])
def test_critic_mlp_one_node_vs_torch(ops, R, T, n_in, H1, H2, E, part):
    """The whole efc-E critic head as ONE autograd node (`_CriticMLP`: reference contextual_sac_value.py:101-107 -> rnn_base.py:461-469 over
    ensemble_linear_model.py:36-49): the head (a2 . w3 + b3) leaves the second GEMM's epilogue (`resel_gemm_f32_head`), the ELU backward and
    bias gradient of the first layer leave the input-gradient GEMM's epilogue (`resel_gemm_f32_dact`) - against plain torch autograd on the
    CPU: q, dx (or only the differentiated column block: the actor step's form), all six parameter gradients.  Row counts that are not
    multiples of 256 exercise the edge tiles of both epilogues; the second size is configs[1]'s critic at 4 rows."""
    from offpolicy_rnn.models.ensemble_linear_model import EnsembleLinear, critic_mlp, critic_mlp_fusable
    torch.manual_seed(5)
    l1, l2, l3 = EnsembleLinear(n_in, H1, E), EnsembleLinear(H1, H2, E, desire_ndim=4), EnsembleLinear(H2, 1, E, desire_ndim=4)
    for l in (l1, l2, l3):
        torch.nn.init.normal_(l.bias, std=0.3)
    x = torch.randn(R, T, n_in)
    w = torch.randn(E, R, T, 1)
    col0, k = n_in // 2, n_in - n_in // 2

    def ref():
        xs = x.clone().requires_grad_(True)
        h = xs
        for i, l in enumerate((l1, l2, l3)):
            h = torch.einsum('...ti,eio->e...to', h, l.weight) if i == 0 else torch.einsum('e...ti,eio->e...to', h, l.weight)
            h = h + l.bias.view(E, 1, 1, -1)
            if i < 2:
                h = torch.nn.functional.elu(h)
        (h * w).sum().backward()
        out = [h.detach(), xs.grad[..., col0:] if part else xs.grad] + [p.grad.clone() for l in (l1, l2, l3) for p in (l.weight, l.bias)]
        for l in (l1, l2, l3):
            l.zero_grad()
        return out

    expect = ref()
    for l in (l1, l2, l3):
        l.cuda()
    elu, ident = torch.nn.ELU(), torch.nn.Identity()
    xs = x.cuda().requires_grad_(not part)
    assert critic_mlp_fusable(l1, elu, l2, elu, l3, ident, xs), 'the fused-epilogue forms must take this shape'
    calls0 = ops.GEMM_FLOPS[0]
    xp = None
    if part:
        xp = xs[..., col0:].clone().requires_grad_(True)
        q = critic_mlp(l1, l2, l3, xs, grad_part=(xp, col0))
    else:
        q = critic_mlp(l1, l2, l3, xs)
    assert q.shape == (E, R, T, 1) and ops.GEMM_FLOPS[0] > calls0
    (q * w.cuda()).sum().backward()
    got = [q, xp.grad if part else xs.grad] + [p.grad for l in (l1, l2, l3) for p in (l.weight, l.bias)]
    names = ['q', 'dx', 'dW1', 'db1', 'dW2', 'db2', 'dW3', 'db3']
    for nm, a, b in zip(names, got, expect):
        close(a, b, rtol=2e-4, atol_scale=5e-5, name=nm)
    # and the node equals the layer-by-layer nodes it replaces (same kernels for the products: tight)
    for l in (l1, l2, l3):
        l.zero_grad()
    from offpolicy_rnn.models.ensemble_linear_model import ensemble_head
    xs2 = x.cuda().requires_grad_(not part)
    xp2 = xs2[..., col0:].clone().requires_grad_(True) if part else None
    h1 = l1(xs2, act='elu', grad_part=(xp2, col0) if part else None)
    q2 = ensemble_head(l2, l3, h1)
    (q2 * w.cuda()).sum().backward()
    got2 = [q2, xp2.grad if part else xs2.grad] + [p.grad for l in (l1, l2, l3) for p in (l.weight, l.bias)]
    for nm, a, b in zip(names, got, got2):
        close(a, b.detach().cpu(), rtol=2e-5, atol_scale=2e-5, name='vs layer-by-layer: ' + nm)


@pytest.mark.parametrize('act', [None, 'elu'])
@pytest.mark.parametrize('rows,n_in,n_out', [(5000, 128, 256), (5000, 96, 64), (300, 128, 256), (5000, 128, 6)])
def test_linear_act_fwd_bwd_long_pass_vs_torch(ops, rows, n_in, n_out, act):
    """`ops.linear_act` (fc layer + activation module, reference rnn_base.py:462-474): output, dx, dW, db against torch autograd;
    long passes run in the hand-written GEMM (forward epilogue, mm_nn, wgrad), short ones and the 6-wide head in the library."""
    g = torch.Generator().manual_seed(rows + n_out)
    x, W, b = rnd(2, rows // 2, n_in, g=g), rnd(n_out, n_in, g=g, scale=n_in ** -0.5), rnd(n_out, g=g, scale=0.3)
    w = rnd(2, rows // 2, n_out, g=g)
    xr, Wr, br = (t.clone().requires_grad_(True) for t in (x, W, b))
    yr = torch.nn.functional.linear(xr, Wr, br)
    yr = torch.nn.functional.elu(yr) if act else yr
    (yr * w).sum().backward()
    xs, Ws, bs = (t.clone().cuda().requires_grad_(True) for t in (x, W, b))
    flops0 = ops.GEMM_FLOPS[0]
    y = ops.linear_act(xs, Ws, bs, act)
    (y * w.cuda()).sum().backward()
    close_fwd(y, yr.detach(), name='y')
    close(xs.grad, xr.grad, name='dx')
    close(Ws.grad, Wr.grad, rtol=2e-4, atol_scale=5e-5, name='dW')
    close(bs.grad, br.grad, rtol=2e-4, atol_scale=5e-5, name='db')
    assert (ops.GEMM_FLOPS[0] > flops0) == (rows >= ops.GEMM_F32_MIN_ROWS and min(n_in, n_out) >= ops.GEMM_F32_MIN_DIM)


def test_linear_act_vs_torch(ops):
    g = torch.Generator().manual_seed(9)
    x, W, b, go = rnd(5, 40, 17, g=g), rnd(128, 17, g=g, scale=0.3), rnd(128, g=g), rnd(5, 40, 128, g=g)
    xr, Wr, br = x.clone().requires_grad_(True), W.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = torch.nn.functional.elu(torch.nn.functional.linear(xr, Wr, br))
    ref.backward(go)
    xc, Wc, bc = (t.cuda().requires_grad_(True) for t in (x, W, b))
    out = ops.linear_act(xc, Wc, bc, 'elu')
    out.backward(go.cuda())
    close_fwd(out, ref, name='out')
    close(xc.grad, xr.grad, name='dx')
    close(Wc.grad, Wr.grad, rtol=2e-4, atol_scale=5e-5, name='dW')
    close(bc.grad, br.grad, rtol=2e-4, atol_scale=5e-5, name='db')


# ------------------------------------------------------------------------------------------ BASELINE-size properties
def test_selective_scan_full_size_properties(ops, monkeypatch):
    """Config-2 shapes (B 64, T' 1043, d_inner 512, N 32: too big for the CPU oracle in a test) through size-independent
    properties of the recurrence: (1) y is linear in u for fixed delta / B / C (no gate), forward and in the u-gradient;
    (2) a `start` reset makes the rest of the row independent of everything before it - the suffix of a packed row
    equals the same suffix processed alone; (3) the last row equals that row processed as a batch of one."""
    g = torch.Generator().manual_seed(21)
    B, L, Di, N = 64, 1043, 512, 32
    dev = 'cuda'
    u1, u2 = rnd(B, L, Di, g=g).to(dev), rnd(B, L, Di, g=g).to(dev)
    delta = (rnd(B, L, Di, g=g) * 0.5).to(dev)
    A = (-torch.exp(rnd(Di, N, g=g) * 0.3)).to(dev)
    Bm, Cm = rnd(B, L, N, g=g).to(dev), rnd(B, L, N, g=g).to(dev)
    D, db = rnd(Di, g=g).to(dev), (rnd(Di, g=g) * 0.1).to(dev)
    start = torch.zeros(B, L, device=dev)
    start[:, 0] = 1
    cut = 517
    start[:, cut] = 1

    def run(u, st=start, sl=slice(None), rows=slice(None)):
        return ops.selective_scan_tm(u[rows, sl].contiguous(), delta[rows, sl].contiguous(), A, Bm[rows, sl].contiguous(),
                                     Cm[rows, sl].contiguous(), D, None, db, st[rows, sl].contiguous(), True)

    y1, y2, y12 = run(u1), run(u2), run(u1 + 2 * u2)
    scale = y12.abs().max().item()
    assert (y12 - (y1 + 2 * y2)).abs().max().item() <= 2e-5 * scale, 'linearity in u'
    tail = run(u1, sl=slice(cut, None))
    assert (tail - y1[:, cut:]).abs().max().item() <= 2e-5 * scale, 'reset isolation / packing'
    # (3) row independence, bit for bit - for the right reason: BOTH sides on the one-pass kernels (a batch of one would otherwise be
    # cut into time segments, whose arithmetic differs from the one-pass scan in the last bits whenever the carried state matters)
    monkeypatch.setattr(ops, 'SSCAN_TIME_SEGMENTS', 1)
    one = run(u1, rows=slice(B - 1, B))
    assert torch.equal(one, y1[B - 1:]), 'row independence must be bit-exact'
    monkeypatch.setattr(ops, 'SSCAN_TIME_SEGMENTS', 0)
    # backward: d/du of sum(w * y) is linear in w
    uu = u1.clone().requires_grad_(True)
    w1, w2 = rnd(B, L, Di, g=g).to(dev), rnd(B, L, Di, g=g).to(dev)
    y = run(uu)
    g1, = torch.autograd.grad(y, uu, w1, retain_graph=True)
    g2, = torch.autograd.grad(y, uu, w2, retain_graph=True)
    g12, = torch.autograd.grad(y, uu, w1 - 3 * w2)
    assert (g12 - (g1 - 3 * g2)).abs().max().item() <= 2e-5 * g12.abs().max().item(), 'backward linearity'
    # bitwise reproducibility of the backward (no atomics anywhere)
    g1b, = torch.autograd.grad(run(uu), uu, w1)
    assert torch.equal(g1, g1b)


def test_conv_full_size_properties(ops):
    """Config-2 conv (B 64, T' 1043, d_inner 512, K 16): causality, mask packing and agreement of the strided (x | z)
    input with a contiguous copy - bit-exact, the arithmetic order does not depend on the layout."""
    g = torch.Generator().manual_seed(22)
    B, L, Di, Kw = 64, 1043, 512, 16
    dev = 'cuda'
    xz = rnd(B, L, 2 * Di, g=g).to(dev)
    w, bias = (rnd(Di, 1, Kw, g=g) * 0.2).to(dev), (rnd(Di, g=g) * 0.1).to(dev)
    mask = (torch.rand(B, L, 1, generator=g) > 0.05).float().to(dev)
    y = ops.causal_conv1d_fn(xz[..., :Di], w, bias, mask, True)
    y2 = ops.causal_conv1d_fn(xz[..., :Di].contiguous(), w, bias, mask, True)
    assert torch.equal(y, y2)
    x3 = xz.clone()
    x3[:, 700:, :Di] += 1.0                            # a change at t >= 700 must not reach outputs before 700
    y3 = ops.causal_conv1d_fn(x3[..., :Di], w, bias, mask, True)
    assert torch.equal(y3[:, :700], y[:, :700]) and not torch.equal(y3[:, 700:], y[:, 700:])
    gap = mask.clone()
    gap[:, 300:300 + Kw] = 0                           # a masked gap of K steps separates the two halves completely
    ya = ops.causal_conv1d_fn(xz[..., :Di], w, bias, gap, True)
    yb = ops.causal_conv1d_fn(xz[:, 300 + Kw:, :Di], w, bias, gap[:, 300 + Kw:], True)
    assert (ya[:, 300 + Kw:] - yb).abs().max().item() <= 1e-6 * ya.abs().max().item()


@pytest.mark.parametrize('K,Wd,Nd,ldn,tr', [(66752, 512, 16, 80, False), (66752, 512, 80, 80, True), (1001, 132, 33, 40, False),
                                            (7, 4, 1, 1, True), (4100, 1024, 96, 96, False)])
def test_atb_long_reduction_gemm(ops, K, Wd, Nd, ldn, tr):
    g = torch.Generator().manual_seed(K + Wd)
    wide = rnd(K, Wd, g=g).cuda()
    narrow_full = rnd(K, ldn, g=g).cuda()
    narrow = narrow_full[:, :Nd]
    got = ops.atb(wide, narrow, tr)
    ref = (wide.double().t() @ narrow.double())
    ref = ref.t() if tr else ref
    close(got.cpu(), ref.float().cpu(), rtol=1e-4, atol_scale=2e-5, name='atb')
    assert torch.equal(got, ops.atb(wide, narrow, tr))           # fixed summation order


def test_encode_concat_padded_block_diagonal_gemm_vs_separate_linears(ops):
    """`policy_value_models/_inputs.encode_concat` (reference contextual_sac_value.py:90-99: cat of per-input Linear encoders): the
    41-column input is zero-padded to 44 so that the hand-written GEMM takes the long pass; output and every gradient against the
    plain `cat([enc(x)])` on the CPU, at a library-sized and a hand-written-GEMM-sized batch."""
    from offpolicy_rnn.policy_value_models._inputs import encode_concat
    torch.manual_seed(4)
    dims = (17, 17, 6, 1)
    for rows in (6, 80):                                   # x 64 tokens: 384 and 5120 tokens
        mods = [torch.nn.Linear(d, 128) for d in dims]
        xs = [torch.randn(rows, 64, d) for d in dims]
        xs[2].requires_grad_(True)                         # the action input carries a gradient in the actor step
        ref = torch.cat([m(x) for m, x in zip(mods, xs)], dim=-1)
        w = torch.randn_like(ref)
        (ref * w).sum().backward()
        expect = [ref.detach(), xs[2].grad.clone()] + [p.grad.clone() for m in mods for p in (m.weight, m.bias)]
        xs[2].grad = None
        for m in mods:
            m.zero_grad()
            m.cuda()
        xc = [x.detach().cuda() for x in xs]
        xc[2].requires_grad_(True)
        flops0 = ops.GEMM_FLOPS[0]
        out = encode_concat(list(zip(mods, xc)))
        (out * w.cuda()).sum().backward()
        got = [out, xc[2].grad] + [p.grad for m in mods for p in (m.weight, m.bias)]
        for i, (a, b) in enumerate(zip(got, expect)):
            close(a, b, rtol=2e-4, atol_scale=5e-5, name=f'rows={rows} tensor {i}')
        assert (ops.GEMM_FLOPS[0] > flops0) == (rows * 64 >= ops.GEMM_F32_MIN_ROWS)


# ------------------------------------------------------------------------------------------ fp32 MFMA GEMM
@pytest.mark.parametrize('M,N,K,akc,bkc,bias,act,batch', [
    (300, 256, 384, True, True, True, 'elu', 1),        # forward with the fused tail, ragged M
    (1000, 132, 72, True, True, False, None, 1),        # ragged N / K tails
    (513, 384, 256, True, False, False, None, 1),       # dgrad: weight read [K][rows]
    (20000, 128, 256, False, False, False, None, 1),    # wgrad: split over the 20 000-long reduction
    (260, 256, 256, True, False, True, 'elu', 8),       # per-member ensemble layer [E, in, out] with bias + ELU
    (17000, 256, 256, False, False, False, None, 8),    # per-member weight gradient (batched split-K)
    (16640, 512, 136, True, True, True, 'elu', 1),      # 520 tiles on 512 block slots: the last 8 are K-split, tail + bias + ELU in the fix-up
    (600, 260, 99, False, False, False, None, 1),       # both operands [K][rows], odd K (guarded last step)
    (600, 260, 100, True, False, True, None, 2),        # mixed layouts, batch of 2, K tail
    (66752, 256, 64, True, True, True, None, 1),        # two K steps per item: the persistent pipeline crosses items every other step
    (5000, 256, 16, True, True, True, None, 1),         # K shorter than one K step (every mode runs the fp32 kernel there)
    (1024, 128, 28, False, True, False, None, 1),
    (256, 512, 8344, False, False, False, None, 1),     # weight gradient whose LAST K slice is shorter than one K step (24 of 32)
    (1024, 256, 4172, False, False, False, None, 1),    # the same with 12 left
    (1500, 200, 4100, True, True, True, 'elu', 1),      # ragged everything, 4 k in the last step
])
@pytest.mark.parametrize('split', [0, 6])
def test_gemm_f32_vs_fp64_product(ops, M, N, K, akc, bkc, bias, act, batch, split):
    """resel_gemm_f32 against an fp64 product: 1e-5 of the largest output magnitude in every product mode (0: fp32 MFMA, exact
    products; 9 / 6: exact three-way bf16 operand split on the bf16 MFMA, all nine / the six leading plane products, operands
    split once per block into bf16 planes in LDS (gemm_bf3.hip); 106: mode 6 on the first-edition kernel); both operand layouts, ragged edges, batch strides, fused bias + ELU, deterministic K split."""
    g = torch.Generator().manual_seed(M + N + K)
    sh = (batch,) if batch > 1 else ()
    A = torch.randn(*sh, *((M, K) if akc else (K, M)), generator=g)
    B = torch.randn(*sh, *((N, K) if bkc else (K, N)), generator=g) / K ** 0.5
    b = torch.randn(*sh, N, generator=g) if bias else None
    Ad = A.double() if akc else A.double().transpose(-1, -2)
    Bd = B.double().transpose(-1, -2) if bkc else B.double()
    ref = Ad @ Bd
    if bias:
        ref = ref + b.double().unsqueeze(-2)
    if act == 'elu':
        ref = torch.nn.functional.elu(ref)
    out = ops.gemm_f32(A.cuda(), B.cuda(), akc, bkc, None if b is None else b.cuda(), act, split=split)
    close(out, ref.float(), rtol=1e-5, atol_scale=1e-6, name='gemm_f32')
    out2 = ops.gemm_f32(A.cuda(), B.cuda(), akc, bkc, None if b is None else b.cuda(), act, split=split)
    assert torch.equal(out, out2)                       # bitwise reproducible (no atomics)


@pytest.mark.parametrize('M,N,K,akc,bkc,bias,act,batch', [
    (47, 6, 256, True, True, True, None, 1),            # the 6-wide TD3 / discrete head of a short pass
    (47, 256, 6, True, False, False, None, 1),          # its input gradient: rows of 24 bytes, a 6-long reduction
    (6, 256, 47, False, False, False, None, 1),         # its weight gradient
    (6, 256, 66752, False, False, False, None, 1),      # ... over all tokens of configs[1]: the reduction is cut over grid.z, partial tiles summed in order
    (1043, 18, 64, True, True, False, None, 1),         # x_proj of smamba_s8: 2 + 2 x 8 output columns
    (1043, 64, 2, True, True, False, 'softplus', 1),    # its rank-2 dt_proj with the softplus epilogue
    (1, 256, 384, True, True, True, 'elu', 1),          # one rollout token against a whole weight matrix: the rows form
    (8, 2048, 256, True, True, True, None, 1),          # eight environments
    (3, 130, 260, True, True, False, 'elu', 2),         # rows form, batch of 2, ragged N
    (5, 7, 3, True, True, True, 'elu', 3),              # everything tiny and odd, batch of 3
    (9, 12, 40, True, False, True, None, 1),            # 9 rows: just past the rows form, [K, N] weight
    (300, 5, 33, False, True, True, 'elu', 2),          # transposed A, odd extents, batch
])
def test_gemm_f32_takes_every_shape(ops, M, N, K, akc, bkc, bias, act, batch):
    """`resel_gemm_f32x` refuses no shape and leaves none to a vendor library (verdict r05 item 7): operands the matrix-core editions cannot
    read - rows that are not 16-byte multiples, reductions shorter than a matrix-instruction step - and the M <= 8 rows of a rollout step
    run csrc/gemm_any.hip (exact fp32 FMAs): against fp64, every epilogue, bitwise reproducible, magnitude published."""
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    sh = (batch,) if batch > 1 else ()
    A = torch.randn(*sh, *((M, K) if akc else (K, M)), generator=g)
    B = torch.randn(*sh, *((N, K) if bkc else (K, N)), generator=g) / K ** 0.5
    b = torch.randn(*sh, N, generator=g) if bias else None
    Ad = A.double() if akc else A.double().transpose(-1, -2)
    Bd = B.double().transpose(-1, -2) if bkc else B.double()
    ref = Ad @ Bd
    if bias:
        ref = ref + b.double().unsqueeze(-2)
    if act == 'elu':
        ref = torch.nn.functional.elu(ref)
    if act == 'softplus':
        ref = torch.nn.functional.softplus(ref)
    kw = dict(bias=None if b is None else b.cuda(), act=act)
    out = ops.gemm_f32(A.cuda(), B.cuda(), akc, bkc, **kw)
    close(out, ref.float(), rtol=1e-5, atol_scale=2e-6, name='gemm_any')
    assert torch.equal(out, ops.gemm_f32(A.cuda(), B.cuda(), akc, bkc, **kw))
    # the accumulating form and an output placed in a wider buffer (row stride != N)
    if act is None and batch == 1:
        wide = torch.randn(M, N + 5, generator=g).cuda()
        want = wide[:, 2:2 + N].double().cpu() + ref
        ops.gemm_f32(A.cuda(), B.cuda(), akc, bkc, kw['bias'], ops.GEMM_ACCUMULATE, out=wide[:, 2:2 + N])
        close(wide[:, 2:2 + N], want.float(), rtol=1e-5, atol_scale=2e-6, name='gemm_any accumulate')
    # magnitude publication: a handle for the output bounds it
    h, epoch = ops.amax_slot(torch.device('cuda'))
    out3 = ops.gemm_f32(A.cuda(), B.cuda(), akc, bkc, amax_out=(h, ops._p(h), epoch), **kw)
    assert float(ops.amax_value(h)) == pytest.approx(float(out3.abs().max()), rel=1e-6)


def test_gemm_f32_random_small_shapes(ops):
    """120 random shapes up to 300 x 300 x 600 in all four operand layouts, with / without bias, ELU and batch: whichever kernel the entry
    picks (rows form, generic tiles, first edition, one-role / producer-consumer editions), the product matches fp64."""
    rs = np.random.RandomState(7)
    g = torch.Generator().manual_seed(7)
    for it in range(120):
        M, N, K = (int(rs.choice([1, 2, 3, 5, 8, 9, 17, 47, 64, 129, 131, 300])), int(rs.choice([1, 3, 4, 6, 12, 18, 64, 130, 256, 300])),
                   int(rs.choice([1, 2, 3, 6, 16, 31, 32, 33, 44, 64, 100, 256, 600])))
        akc, bkc, batch = bool(rs.randint(2)), bool(rs.randint(2)), int(rs.choice([1, 1, 2, 3]))
        bias, act = bool(rs.randint(2)), [None, 'elu'][rs.randint(2)]
        sh = (batch,) if batch > 1 else ()
        A = torch.randn(*sh, *((M, K) if akc else (K, M)), generator=g)
        B = torch.randn(*sh, *((N, K) if bkc else (K, N)), generator=g) / K ** 0.5
        b = torch.randn(*sh, N, generator=g) if bias else None
        ref = (A.double() if akc else A.double().transpose(-1, -2)) @ (B.double().transpose(-1, -2) if bkc else B.double())
        if bias:
            ref = ref + b.double().unsqueeze(-2)
        if act == 'elu':
            ref = torch.nn.functional.elu(ref)
        out = ops.gemm_f32(A.cuda(), B.cuda(), akc, bkc, None if b is None else b.cuda(), act)
        close(out, ref.float(), rtol=1e-5, atol_scale=2e-6, name=f'gemm_f32 #{it} M{M} N{N} K{K} akc{akc} bkc{bkc} batch{batch} bias{bias} act{act}')


@pytest.mark.parametrize('M,N,K,x_bf,out', [(1, 768, 256, False, 'bf16'), (4, 256, 256, True, 'bf16'), (8, 256, 256, True, 'round'),
                                             (2, 256, 1024, False, 'f32'), (47, 96, 32, False, 'bf16'), (5, 12, 6, False, 'round')])
def test_gemm_bf16_takes_decode_rows_and_odd_shapes(ops, M, N, K, x_bf, out):
    """The bf16-autocast projections of the cgpt MHA for the few rows of a decode step (rows form, bf16 or fp32 x) and for shapes the
    matrix-core kernel cannot read: operands and bias rounded to bf16, fp32 accumulation, the result rounded as asked - against fp64 on
    the rounded operands."""
    g = torch.Generator().manual_seed(M + N + K)
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
    bf = torch.bfloat16
    ref = x.to(bf).double() @ w.to(bf).double().t() + b.to(bf).double()
    xin = x.to(bf).cuda() if x_bf else x.cuda()
    y = ops.gemm_bf16(xin, w.cuda(), True, True, b.cuda(), bf if out == 'bf16' else torch.float32, round_out=(out == 'round'))
    assert y.dtype == (bf if out == 'bf16' else torch.float32)
    want = ref.float() if out == 'f32' else ref.to(bf).float()
    tol = 1e-5 if out == 'f32' else 1e-2                # one bf16 ulp where the fp32 sums of kernel and reference straddle a rounding boundary
    close(y.float(), want, rtol=tol, atol_scale=tol, name='gemm_bf16 small')
    if out == 'round':
        assert torch.equal(y, y.to(bf).float())         # the fp32 output holds bf16 values


@pytest.mark.parametrize('M,N,K,split', [(66752, 512, 80, 6), (1000, 132, 72, 6), (5000, 256, 64, 0)])
def test_gemm_f32_accumulating_epilogue(ops, M, N, K, split):
    """Epilogue code 2: C += A B (the accumulating form `dxc.addmm_(dx_dbl, x_proj.weight)` of the Mamba mixer's backward), whole
    tiles and K-split tails alike, against the fp64 sum."""
    g = torch.Generator().manual_seed(M + K)
    A, B = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g) / K ** 0.5
    C0 = torch.randn(M, N, generator=g)
    ref = C0.double() + A.double() @ B.double()
    out = C0.clone().cuda()
    ops.gemm_f32(A.cuda(), B.cuda(), True, False, None, ops.GEMM_ACCUMULATE, out=out, split=split)
    close(out, ref.float(), rtol=1e-5, atol_scale=1e-6, name='C += A B')


def test_gemm_f32_split_modes_error_against_fp64(ops):
    """The bf16-split product modes are as accurate as the fp32 instruction: mean and maximum error against an fp64 product, on
    operands with a wide dynamic range (6 decades), K = 4096.  Mode 6 adds at most one product rounding per term to the accumulation's
    error; mode 2: see the next test."""
    g = torch.Generator().manual_seed(5)
    M, N, K = 512, 384, 4096
    A = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, K, generator=g) * 2.0)).cuda()
    B = (torch.randn(N, K, generator=g) * torch.exp(torch.randn(N, K, generator=g) * 2.0)).cuda()
    ref = A.double() @ B.double().t()
    scale = (A.double().abs() @ B.double().abs().t())          # sum |a b|: the natural error scale of a dot product
    err = {}
    for split in (0, 6, 2):
        e = ((ops.gemm_f32(A, B, True, True, split=split).double() - ref).abs() / scale)
        err[split] = (e.mean().item(), e.max().item())
    lib = (((A @ B.t()).double() - ref).abs() / scale)
    print('relative to sum|ab|: mean / max', err, 'library fp32 GEMM', (lib.mean().item(), lib.max().item()))
    for split in (6, 2):
        assert err[split][0] <= 1.5 * err[0][0] + 1e-9 and err[split][1] <= 2.0 * err[0][1] + 1e-9, err
    assert err[6][1] < 1e-6 and err[2][1] < 1e-6


@pytest.mark.parametrize('decades,K', [(2.0, 64), (4.0, 64), (4.0, 256), (4.0, 4096), (0.0, 1024)])
def test_gemm_f32_f16x3_mode_error_against_fp64(ops, decades, K):
    """Mode 2 (fp16 planes of the SCALED operands, three plane products, one accumulator): as accurate as the fp32 instruction
    (mode 0) against fp64 - mean <= 1.5 x, max <= 2 x of its error relative to sum|a b| - from short to long K, with A spanning up
    to 12 decades (randn * exp(4 randn): its residual plane is normal down to 2^-29 max|A|) and B up to 6 decades (its residual
    plane down to 2^-18 max|B|: the operand slot of the weights).  The operand magnitudes come from resel_amax inside the call."""
    g = torch.Generator().manual_seed(int(decades * 10) + K)
    M, N = 512, 384
    A = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, K, generator=g) * decades)).cuda()
    B = (torch.randn(N, K, generator=g) * torch.exp(torch.randn(N, K, generator=g) * min(decades, 2.0))).cuda()
    ref = A.double() @ B.double().t()
    scale = (A.double().abs() @ B.double().abs().t())
    err = {}
    for split in (0, 2, 6):
        e = ((ops.gemm_f32(A, B, True, True, split=split).double() - ref).abs() / scale)
        err[split] = (e.mean().item(), e.max().item())
    print('f16x3 (2) vs fp32 instruction (0) vs bf16 split (6), relative to sum|ab| (mean, max):', err)
    assert err[2][0] <= 1.5 * err[0][0] + 1e-9 and err[2][1] <= 2.0 * err[0][1] + 1e-9, err
    # layouts, bias + ELU epilogue, batch, K tail: against mode 6 at fp32 rounding level
    A3, B3, b3 = torch.randn(3, 700, 200, generator=g).cuda(), torch.randn(3, 200, 264, generator=g).cuda() / 14, torch.randn(3, 264, generator=g).cuda()
    o6 = ops.gemm_f32(A3, B3, True, False, b3, 'elu', split=6)
    o2 = ops.gemm_f32(A3, B3, True, False, b3, 'elu', split=2)
    close(o2, o6.cpu(), rtol=2e-6, atol_scale=1e-6, name='mode 2 vs mode 6, batched + bias + elu')
    Z = torch.zeros(300, 64).cuda()
    assert ops.gemm_f32(Z, torch.randn(128, 64, generator=g).cuda(), True, True, split=2).abs().max().item() == 0.0   # amax = 0


@pytest.mark.parametrize('M,N,K,akc,bkc,bias,act,batch', [
    (300, 256, 384, True, True, True, 'elu', 1), (513, 384, 256, True, False, False, None, 1), (20000, 256, 256, False, False, False, None, 1),
    (16640, 512, 136, True, True, True, 'elu', 1), (600, 260, 99, False, False, False, None, 1), (260, 256, 256, True, False, True, 'elu', 8),
    (1500, 200, 4100, True, True, True, 'elu', 1)])
def test_gemm_f32_two_plane_mode(ops, M, N, K, akc, bkc, bias, act, batch):
    """Mode 3 ("bf16x3": two bf16 planes per operand, three plane products - torch's float32 matmul precision 'high'): every dropped
    term is <= 2^-16 |a b|, so the error against fp64 is bounded by 3 x 2^-16 sum |a b| (measured ~ 2^-18 of the output scale);
    same layouts / tails / fix-up as the fp32-accurate modes, bitwise reproducible."""
    g = torch.Generator().manual_seed(M + N + K)
    sh = (batch,) if batch > 1 else ()
    A = torch.randn(*sh, *((M, K) if akc else (K, M)), generator=g)
    B = torch.randn(*sh, *((N, K) if bkc else (K, N)), generator=g) / K ** 0.5
    b = torch.randn(*sh, N, generator=g) if bias else None
    Ad = A.double() if akc else A.double().transpose(-1, -2)
    Bd = B.double().transpose(-1, -2) if bkc else B.double()
    ref = Ad @ Bd
    bound = 3 * 2.0 ** -16 * (Ad.abs() @ Bd.abs()) + 1e-6
    if bias:
        ref = ref + b.double().unsqueeze(-2)
    out = ops.gemm_f32(A.cuda(), B.cuda(), akc, bkc, None if b is None else b.cuda(), None, split=3)
    assert ((out.double().cpu() - ref).abs() <= bound).all()
    if act == 'elu':
        out_a = ops.gemm_f32(A.cuda(), B.cuda(), akc, bkc, None if b is None else b.cuda(), act, split=3)
        close(out_a, torch.nn.functional.elu(ref).float(), rtol=1e-4, atol_scale=1e-5, name='gemm_f32 mode 3 + elu')
    out2 = ops.gemm_f32(A.cuda(), B.cuda(), akc, bkc, None if b is None else b.cuda(), None, split=3)
    assert torch.equal(out, out2)


def test_gemm_f32_follows_torch_matmul_precision(ops, monkeypatch):
    """No RESEL_GEMM_SPLIT: 'highest' (torch's default and the reference's setting) = the fp32-accurate modes - 2 where the operand
    magnitudes are known (a handle from the producer, or a pre-pass that is cheap next to the GEMM), 6 otherwise; 'high' = mode 3."""
    monkeypatch.setattr(ops, 'GEMM_SPLIT', None)
    g = torch.Generator().manual_seed(1)
    A, B = torch.randn(1024, 512, generator=g).cuda(), torch.randn(256, 512, generator=g).cuda()
    keep = torch.get_float32_matmul_precision()
    try:
        torch.set_float32_matmul_precision('highest')
        assert ops.gemm_split() == 2
        assert torch.equal(ops.gemm_f32(A, B), ops.gemm_f32(A, B, split=6))          # small, untagged operands: a pre-pass does not pay - mode 6
        ha, hb = ops.amax(A), ops.amax(B)
        assert abs(ops.amax_value(ha) - A.abs().max().item()) == 0.0 and abs(ops.amax_value(hb) - B.abs().max().item()) == 0.0
        c2 = ops.gemm_f32(A, B, amax_a=ha, amax_b=hb)                                  # handles at hand: mode 2
        assert torch.equal(c2, ops.gemm_f32(A, B, split=2, amax_a=ha, amax_b=hb)) and not torch.equal(c2, ops.gemm_f32(A, B, split=6))
        assert (c2 - ops.gemm_f32(A, B, split=6)).abs().max().item() <= 2e-6 * 512 ** 0.5 * 16
        big, wide = torch.randn(20000, 512, generator=torch.Generator().manual_seed(2)).cuda(), torch.randn(2048, 512, generator=g).cuda()
        out = ops.gemm_f32(big, wide)                                                  # a long, wide pass: pre-pass + mode 2, output tagged with its magnitude
        assert ops.amax_of(big) is not None and abs(ops.amax_value(ops.amax_of(out)) - out.abs().max().item()) == 0.0
        torch.set_float32_matmul_precision('high')
        assert ops.gemm_split() == 3 and torch.equal(ops.gemm_f32(A, B), ops.gemm_f32(A, B, split=3))
        assert not torch.equal(ops.gemm_f32(A, B, split=3), ops.gemm_f32(A, B, split=6))
    finally:
        torch.set_float32_matmul_precision(keep)


# ------------------------------------------------------------------------------------------ time-parallel selective scan
@pytest.mark.parametrize('B,L,Di,N,segs', [(2, 333, 128, 32, 4), (1, 200, 64, 16, 3), (3, 1043, 64, 32, 0), (2, 97, 64, 8, 2)])
def test_selective_scan_time_segments_equal_the_one_pass_scan(ops, monkeypatch, B, L, Di, N, segs):
    """Small batches are cut into time segments scanned in parallel (local pass, carry of the segment states, final pass; the
    backward: local adjoint pass, reverse carry, full pass per segment).  Forward, every gradient and the oracle agree;
    resets fall inside and on the edges of segments; segs = 0 lets the library choose (it splits at these sizes)."""
    g = torch.Generator().manual_seed(B * L + N)
    u, delta, z = rnd(B, L, Di, g=g), rnd(B, L, Di, g=g, scale=0.5), rnd(B, L, Di, g=g)
    Bm, Cm = rnd(B, L, N, g=g), rnd(B, L, N, g=g)
    A, D, db = -torch.exp(rnd(Di, N, g=g, scale=0.3)), rnd(Di, g=g), rnd(Di, g=g, scale=0.1)
    start = make_start(B, L, g, p=0.02)
    start[0, 32] = 1
    start[-1, min(L - 1, 64)] = 1
    w = rnd(B, L, Di, g=g)
    outs = []
    for mode in (1, segs):
        monkeypatch.setattr(ops, 'SSCAN_TIME_SEGMENTS', mode)
        ins = [t.clone().cuda().requires_grad_(True) for t in (u, delta, A, Bm, Cm, D, z, db)]
        out, last = ops.selective_scan_tm(*ins, start.cuda(), True, return_last_state=True)
        (out * w.cuda()).sum().backward()
        outs.append((out.detach(), last.detach(), [t.grad for t in ins]))
    (o1, l1, g1), (o2, l2, g2) = outs
    close(o2, o1.cpu(), rtol=1e-5, atol_scale=1e-6, name='out')
    close(l2, l1.cpu(), rtol=1e-5, atol_scale=1e-6, name='last_state')
    for a, b, nm in zip(g2, g1, ('du', 'ddelta', 'dA', 'dB', 'dC', 'dD', 'dz', 'dbias')):
        close(a, b.cpu(), rtol=1e-4, atol_scale=1e-5, name=nm)
    ref_in = [t.clone().requires_grad_(True) for t in (u, delta, A, Bm, Cm, D, z, db)]
    ref, _ = K.selective_scan_ref(*ref_in, start, True)
    (ref * w).sum().backward()
    close_fwd(o2, ref, name='out vs oracle')
    for a, b, nm in zip(g2, ref_in, ('du', 'ddelta', 'dA', 'dB', 'dC', 'dD', 'dz', 'dbias')):
        close(a, b.grad, rtol=2e-4, atol_scale=5e-5, name=nm + ' vs oracle')


# ------------------------------------------------------------------------------------------ mixed-precision GEMM (cgpt projections)
@pytest.mark.parametrize('M,N,K,akc,bkc,a_bf,b_bf,c_bf,bias', [
    (1000, 768, 256, True, True, False, False, True, True),        # Wqkv forward: fp32 activations, fp32 weight -> bf16
    (1000, 256, 256, True, True, True, False, False, True),        # out_proj forward: bf16 activations -> fp32
    (1000, 256, 768, True, False, True, False, False, False),      # Wqkv dgrad: bf16 gradient, weight [K][rows] -> fp32
    (1000, 256, 256, True, False, False, False, True, False),      # out_proj dgrad -> bf16
    (768, 256, 5000, False, False, True, False, False, False),     # Wqkv wgrad: K = tokens, split over blocks
    (256, 256, 33000, False, False, False, True, False, False),    # out_proj wgrad: bf16 activations as the [K][rows] 128-row operand
    (300, 132, 100, True, True, False, False, True, True),         # ragged M / N, K tail
    (4100, 260, 36, False, False, True, True, False, False),       # both bf16 [K][rows], K tail, ragged N
    (66000, 128, 64, True, True, False, False, False, True),       # two K steps per item, more than two rounds of tiles
])
def test_gemm_bf16_vs_rounded_operands_in_fp64(ops, M, N, K, akc, bkc, a_bf, b_bf, c_bf, bias):
    """resel_gemm_bf16: C = bf16(A) (.) bf16(B) + bf16(bias) with fp32 accumulation.  Reference: the same rounded operands multiplied
    in fp64 (a bf16 x bf16 product is exact in fp32, so only the accumulation order differs): 1e-5 of the largest output for fp32
    results, one bf16 rounding (2^-8) on top for bf16 results; bitwise reproducible."""
    g = torch.Generator().manual_seed(M + N + K)
    bf = torch.bfloat16
    A = torch.randn((M, K) if akc else (K, M), generator=g)
    B = torch.randn((N, K) if bkc else (K, N), generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) if bias else None
    A_in = A.to(bf) if a_bf else A
    B_in = B.to(bf) if b_bf else B
    Ar, Br = A.to(bf).double(), B.to(bf).double()
    ref = (Ar if akc else Ar.t()) @ (Br.t() if bkc else Br)
    if bias:
        ref = ref + b.to(bf).double()
    out = ops.gemm_bf16(A_in.cuda(), B_in.cuda(), akc, bkc, None if b is None else b.cuda(), bf if c_bf else torch.float32)
    assert out.dtype == (bf if c_bf else torch.float32) and out.shape == (M, N)
    close(out, ref.float(), rtol=(4e-3 if c_bf else 1e-5), atol_scale=(1e-3 if c_bf else 1e-6), name='gemm_bf16')
    out2 = ops.gemm_bf16(A_in.cuda(), B_in.cuda(), akc, bkc, None if b is None else b.cuda(), bf if c_bf else torch.float32)
    assert torch.equal(out, out2)


@pytest.mark.parametrize('x_bf,out_bf', [(False, True), (True, False)])
def test_linear_bf16_node_matches_the_autocast_graph(ops, x_bf, out_bf):
    """The one-node form of F.linear under bf16 autocast (operands cast to bf16, fp32 master parameters) against that graph built
    from torch ops: forward, input gradient, weight and bias gradients at bf16 tolerance."""
    g = torch.Generator().manual_seed(7)
    bf = torch.bfloat16
    T, K, N = 2052, 256, 768 if not x_bf else 256
    x = torch.randn(T, K, generator=g)
    w, b = torch.randn(N, K, generator=g) / 16, torch.randn(N, generator=g) * 0.1
    dy = torch.randn(T, N, generator=g)

    def run(fn):
        xs = (x.to(bf) if x_bf else x).cuda().requires_grad_(True)
        ws, bs = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
        y = fn(xs, ws, bs)
        (y.float() * dy.cuda()).sum().backward()
        return y, xs.grad, ws.grad, bs.grad

    y_ref, dx_ref, dw_ref, db_ref = run(lambda xs, ws, bs: torch.nn.functional.linear(xs.to(bf), ws.to(bf), bs.to(bf)))
    y, dx, dw, db = run(lambda xs, ws, bs: ops.linear_bf16(xs, ws, bs, bf if out_bf else torch.float32))
    assert y.dtype == (bf if out_bf else torch.float32) and dx.dtype == (bf if x_bf else torch.float32) and dw.dtype == torch.float32
    close(y, y_ref.float().cpu(), rtol=1e-2, atol_scale=2e-3, name='y')
    close(dx, dx_ref.float().cpu(), rtol=1e-2, atol_scale=2e-3, name='dx')
    close(dw, dw_ref.cpu(), rtol=1e-2, atol_scale=2e-3, name='dw')
    close(db, db_ref.cpu(), rtol=1e-2, atol_scale=2e-3, name='db')


@pytest.mark.parametrize('K,N,act', [(17, 128, None), (256, 6, None), (6, 6, 'elu'), (41, 132, 'elu'), (256, 12, None)])
def test_linear_act_pads_odd_widths_on_long_passes(ops, K, N, act):
    """Input / output widths that are not multiples of 4 (17-wide observation and 6-wide action encoders, the 6-wide TD3 action
    head) are zero-padded inside the LinearAct node on long passes, so that forward, input gradient and weight gradient all run on the
    hand-written GEMM; values and gradients equal F.linear (+ ELU)."""
    g = torch.Generator().manual_seed(K * 7 + N)
    T = 4500
    x, w, b = torch.randn(3, T // 3, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g) * 0.1
    dy = torch.randn(3, T // 3, N, generator=g)

    def run(dev, fn):
        ts = [t.clone().to(dev).requires_grad_(True) for t in (x, w, b)]
        y = fn(*ts)
        (y * dy.to(dev)).sum().backward()
        return y, [t.grad for t in ts]

    ref_fn = lambda x_, w_, b_: (torch.nn.functional.elu if act else (lambda t: t))(torch.nn.functional.linear(x_, w_, b_))
    y_ref, g_ref = run('cpu', ref_fn)
    y, gr = run('cuda', lambda x_, w_, b_: ops.linear_act(x_, w_, b_, act))
    assert y.shape == y_ref.shape
    close_fwd(y, y_ref, name='y')
    for nm, a, b_ in zip(('dx', 'dw', 'db'), gr, g_ref):
        assert a.shape == b_.shape
        close(a, b_, rtol=2e-4, atol_scale=5e-5, name=nm)


# ------------------------------------------------------------------------------------------ masked losses
@pytest.mark.parametrize('E,shape', [(8, (5, 37, 1)), (2, (3, 1043, 1)), (1, (2, 9, 1))])
def test_masked_losses_vs_torch_autograd(ops, E, shape):
    """`resel_q_loss_*` / `resel_actor_loss_*` (reference sac_full_length_rnn_ensembleQ.py:80-81,105-114, sac_full_length_rnn_redq.py:37-49)
    against the same sums spelt in torch autograd on the CPU: values at 1e-6 of their scale, gradients element-wise at 1e-6."""
    g = torch.Generator().manual_seed(E + shape[1])
    q = torch.randn(E, *shape, generator=g)
    y = torch.randn(*shape, generator=g)
    mask = (torch.rand(*shape, generator=g) > 0.3).float()
    logp = torch.randn(*shape, generator=g)
    la = torch.tensor([-0.7])
    qr = q.clone().requires_grad_(True)
    ref = ((qr - y.unsqueeze(0)).pow(2).sum(dim=0) * mask).sum()
    (ref * 1.5).backward()
    qc = q.cuda().requires_grad_(True)
    got = ops.masked_q_loss(qc, y.cuda(), mask.cuda())
    (got * 1.5).backward()
    assert abs(got.item() - ref.item()) <= 1e-5 * abs(ref.item())
    close(qc.grad, qr.grad, rtol=1e-6, atol_scale=1e-7, name='dq (critic loss)')
    for reduce_min in (False, True):
        qr, lr = q.clone().requires_grad_(True), logp.clone().requires_grad_(True)
        red = qr.min(dim=0).values if reduce_min else qr.mean(dim=0)
        ref = ((la.exp() * lr - red) * mask).sum()
        ref.backward()
        qc, lc = q.cuda().requires_grad_(True), logp.cuda().requires_grad_(True)
        got, lps = ops.masked_actor_loss(lc, qc, mask.cuda(), la.cuda(), True, reduce_min)
        got.backward()
        assert abs(got.item() - ref.item()) <= 1e-5 * max(1.0, abs(ref.item()))
        assert abs(lps.item() - (logp * mask).sum().item()) <= 1e-5 * max(1.0, (logp * mask).abs().sum().item())
        close(qc.grad, qr.grad, rtol=1e-6, atol_scale=1e-7, name=f'dq (actor loss, min={reduce_min})')
        close(lc.grad, lr.grad, rtol=1e-6, atol_scale=1e-7, name='dlogp')


def test_producers_publish_operand_magnitudes(ops, monkeypatch):
    """With product mode 2 in force the producing kernels leave a magnitude handle on their outputs (include/resel_hip.h "magnitude
    handles"): exact maxima for GEMM / bias_act_bwd / gelu_dropout outputs, the analytic bound sqrt(C) max|w| + max|b| for the norms."""
    monkeypatch.setattr(ops, 'GEMM_SPLIT', 2)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(5000, 256, generator=g).cuda()
    w, b = (1 + 0.1 * torch.randn(256, generator=g)).cuda(), (0.1 * torch.randn(256, generator=g)).cuda()
    y = ops.layer_norm_fn(x, w, b, eps=1e-5)
    bound = ops.amax_value(ops.amax_of(y))
    assert y.abs().max().item() <= bound <= 16 * w.abs().max().item() + b.abs().max().item() + 1e-4
    h = ops.gelu_dropout(x, 0.1)
    assert ops.amax_value(ops.amax_of(h)) == h.abs().max().item()
    gy, _ = ops.bias_act_bwd(x, torch.nn.functional.elu(x), x.shape[0], 'elu', True)
    assert ops.amax_value(ops.amax_of(gy)) == gy.abs().max().item()
    W = torch.randn(2048, 256, generator=g).cuda() / 16
    out = ops.gemm_f32(y, W, True, True)                  # operand magnitude from the norm's handle: no pre-pass, mode 2
    assert ops.LAST_SPLIT[0] == 2 and ops.amax_value(ops.amax_of(out)) == out.abs().max().item()
    ref = y.double() @ W.double().t()
    assert ((out.double() - ref).abs() / (y.double().abs() @ W.double().abs().t())).max().item() < 5e-7


def test_flat_store_publishes_weight_magnitudes_in_one_launch(ops, monkeypatch):
    """`FlatParameterStore.amax_handle`: one `resel_amax_segments` launch covers every tensor of the buffer; the handles follow the buffer -
    refreshed after this library's in-place kernels (ops.PARAM_EPOCH) and after torch-visible in-place writes (`_version`)."""
    from collections import OrderedDict
    from offpolicy_rnn.models.flat_params import FlatParameterStore
    monkeypatch.setattr(ops, 'GEMM_SPLIT', 2)
    torch.manual_seed(0)
    mods = OrderedDict(a=torch.nn.Linear(64, 32), b=torch.nn.Linear(32, 8)).copy()
    for m in mods.values():
        m.cuda()
    st = FlatParameterStore(mods)
    ps = [p for p, _, _ in st.slices]
    for p in ps:
        assert ops.amax_value(ops.weight_amax(p)) == p.detach().abs().max().item()
    h0 = ops.weight_amax(ps[0])
    with torch.no_grad():
        ps[2].mul_(3.0)                                   # torch-visible write to one tensor: that tensor's lookup refreshes the table
    assert ops.amax_value(ops.weight_amax(ps[2])) == ps[2].detach().abs().max().item()
    ops.soft_update_(st.flat, st.flat * 2, 0.5)           # this library's kernel: PARAM_EPOCH moves, next lookup refreshes
    assert ops.amax_value(ops.weight_amax(ps[0])) == ps[0].detach().abs().max().item() and ops.weight_amax(ps[0]) is h0


@pytest.mark.gpu
@pytest.mark.parametrize('M,N,pad', [(32832, 256, 0), (1000, 768, 8), (257, 1024, 0), (4099, 8, 16)])
def test_colsum_bf16_matches_fp64_sum(ops, M, N, pad):
    """`resel_colsum_bf16` (bias gradient of the bf16 projections) against the fp64 column sums of the same bf16 values: fp32
    accumulation error only; repeated calls are bitwise equal (fixed summation order)."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    g = torch.Generator(device='cuda').manual_seed(M + N)
    x = (torch.randn(M, N + pad, device='cuda', generator=g) * 3).to(torch.bfloat16)[:, :N]
    got = ops.colsum_bf16(x)
    want = x.double().sum(0)
    scale = x.double().abs().sum(0)
    assert got.dtype == torch.float32 and got.shape == (N,)
    assert ((got.double() - want).abs() <= 2e-6 * scale + 1e-30).all()
    assert torch.equal(got, ops.colsum_bf16(x))
