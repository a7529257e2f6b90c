"""cgpt dropout (BASELINE configs[2]: `cgpt_h8_l6_p0.1_ml1024_rms`): the counter-keyed keep masks of the attention kernels and
of the element-wise dropout kernel against their oracle restatement.

Reference: `MHA(dropout=dropout)` + `nn.Dropout(dropout)` (offpolicy_rnn/models/flash_attention/TransformerFlashAttention.py:
48,52,67-72,76-85), active in `.train()` passes.  The reference's random streams (flash-attn's and ATen's Philox) are not
reproducible outside those libraries; what is compared is (a) the mask function, bit for bit, (b) the arithmetic around it
(keep / rescale, statistics before the mask, backward through the same mask) at the bf16 tolerance of north_star (1e-2).
Attention oracle: parity unpinned (flash_attn absent)."""
import numpy as np
import pytest
import torch

from oracle import kernels as K
from oracle import network as NW

CFG2 = 'cgpt_h8_l6_p0.1_ml1024_rms'             # BASELINE.json configs[2], literally


# ------------------------------------------------------------------------------------------------ CPU: the counter functions
def test_attention_keep_mask_statistics():
    """Keep rate = (floor(0.9 * 255) + 1) / 256 per head and per byte lane; masks of different offsets / heads / query
    tokens are uncorrelated; p = 0 keeps everything."""
    H, n, p = 8, 256, 0.1
    keep = K.attn_dropout_keep(1234, 8, H, 77, n, p).numpy()
    want = (int(np.floor(0.9 * 255)) + 1) / 256
    assert abs(keep.mean() - want) < 2e-3
    for h in range(H):
        assert abs(keep[h].mean() - want) < 6e-3
    for b in range(4):
        assert abs(keep[:, :, b::4].mean() - want) < 4e-3
    other = K.attn_dropout_keep(1234, 12, H, 77, n, p).numpy()
    both = (keep & other).mean()
    assert abs(both - want * want) < 3e-3                       # independent draws
    a, b = keep[0].astype(np.float64) - want, keep[1].astype(np.float64) - want
    assert abs((a * b).mean()) < 2e-3                           # heads
    a, b = keep[:, :-1].astype(np.float64) - want, keep[:, 1:].astype(np.float64) - want
    assert abs((a * b).mean()) < 2e-3                           # neighbouring queries
    a, b = keep[:, :, :-1].astype(np.float64) - want, keep[:, :, 1:].astype(np.float64) - want
    assert abs((a * b).mean()) < 2e-3                           # neighbouring keys (bytes of one word included)
    assert K.attn_dropout_keep(1, 0, 2, 0, 40, 0.0).all()
    # the mask of a sequence depends on the packed index of its first token, not on what precedes it
    assert torch.equal(K.attn_dropout_keep(5, 4, 2, 300, 33, 0.3), K.attn_dropout_keep(5, 4, 2, 300, 64, 0.3)[:, :33, :33])


def test_elementwise_keep_mask_statistics():
    n, p = 1 << 18, 0.1
    k = K.dropout_keep(99, 4, n, p).numpy()
    assert abs(k.mean() - 0.9) < 2e-3
    assert abs(k[0::2].mean() - 0.9) < 3e-3 and abs(k[1::2].mean() - 0.9) < 3e-3
    d = k.astype(np.float64) - 0.9
    assert abs((d[:-1] * d[1:]).mean()) < 1e-3
    k2 = K.dropout_keep(99, 8, n, p).numpy()
    assert abs((k & k2).mean() - 0.81) < 3e-3
    x = torch.randn(1000)
    y = K.dropout_ref(x, 0.25, 3, 0)
    kept = K.dropout_keep(3, 0, 1000, 0.25)
    assert torch.equal(y[~kept], torch.zeros((~kept).sum())) and torch.allclose(y[kept], x[kept] / 0.75)
    assert K.dropout_ref(x, 0.0, 3, 0) is x


def test_expected_attention_output_is_unbiased_up_to_the_8bit_threshold():
    """E[dropout(P) V] = P V * (keep rate / (1 - p)) - flash-attn's 8-bit threshold keeps 230/256 at p = 0.1."""
    H, n, hd, p = 2, 24, 8, 0.1
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn(n, H, hd, generator=g) for _ in range(3))
    cu = torch.tensor([0, n], dtype=torch.int32)
    base = K.attention_alibi_varlen_ref(q, k, v, cu, K.alibi_slopes(H))
    acc = torch.zeros_like(base)
    R = 400
    for r in range(R):
        acc += K.attention_alibi_varlen_ref(q, k, v, cu, K.alibi_slopes(H), None, p, 7, 4 * r)
    ratio = (230 / 256) / 0.9
    err = (acc / R - base * ratio).abs().max().item()
    assert err < 0.08 * base.abs().max().item()


def test_module_in_training_mode_draws_like_the_oracle(oracle_ops):
    """Host logic: the product's decoder in .train() mode (kernels swapped for the oracle ops) consumes one counter draw per
    dropout site in the order attention, post-attention, FFN hidden, post-FFN - the oracle's `cgpt_layer` with a DropCounter
    at the same start reproduces it exactly; .eval() is the p = 0 function."""
    import oracle_backend
    from offpolicy_rnn.models.rnn_base import RNNBase
    from offpolicy_rnn.models.flash_attention.TransformerFlashAttention import PackedSeqs
    D, lid = 64, 'cgpt_h2_l2_p0.1_ml64_rms'
    torch.manual_seed(2)
    net = RNNBase(D, D, [], ['linear'], [lid])
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    x = torch.randn(2, 21, D)
    table = np.zeros((2, 21), dtype=np.int64)
    table[0, :3], table[1, :1] = (1, 7, 13), (21,)
    spec = dict(layer_type=[lid], activation=['linear'])

    def product(train):
        net.train(train)
        hid = net.make_init_state(2, torch.device('cpu'))
        hid.set_attention_concat_mask(PackedSeqs(table, 21, torch.device('cpu')))
        oracle_backend._host_counter.seed, oracle_backend._host_counter.offset = 11, 40
        return net.meta_forward(x, hid)[0]

    ref_train = NW.rnn_base_forward(sd, spec, x, NW.Flags(seqlens=torch.from_numpy(table), dropout=K.DropCounter(11, 40)))
    ref_eval = NW.rnn_base_forward(sd, spec, x, NW.Flags(seqlens=torch.from_numpy(table)))
    got_train, got_eval = product(True), product(False)
    assert oracle_backend._host_counter.offset == 40             # eval mode draws nothing
    assert torch.allclose(got_train, ref_train, atol=1e-5) and torch.allclose(got_eval, ref_eval, atol=1e-5)
    assert (ref_train - ref_eval).abs().max() > 1e-2             # and the masks do something


# ------------------------------------------------------------------------------------------------ GPU
gpu = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from offpolicy_rnn.hip import ops as o
    return o


def _close(got, ref, tol, name):
    got, ref = got.detach().float().cpu(), ref.detach().float()
    scale = max(ref.abs().max().item(), 1e-6)
    err = (got - ref).abs().max().item()
    print(f'MEASURED cgpt {name}: max err / max|ref| = {err / scale:.3e} (bound {tol:g})')
    assert torch.isfinite(got).all() and err <= tol * scale, f'{name}: max err {err:.3e} vs scale {scale:.3e}'


@gpu
@pytest.mark.parametrize('H,hd,lens,p', [(8, 32, [1, 130, 37, 64], 0.1), (4, 64, [200, 33], 0.1), (8, 32, [1027], 0.1), (2, 32, [70, 300], 0.5)])
def test_attention_dropout_fwd_bwd_vs_oracle_with_the_shared_mask(ops, H, hd, lens, p):
    """Forward, dQ and dK/dV kernels regenerate the same mask as the oracle: out and dqkv at the bf16 tolerance (1e-2 / 2e-2)."""
    g = torch.Generator().manual_seed(sum(lens) + H)
    T = sum(lens)
    qkv = (torch.randn(T, 3, H, hd, generator=g) * 0.8).to(torch.bfloat16)
    dout = torch.randn(T, H, hd, generator=g).to(torch.bfloat16)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
    slopes = K.alibi_slopes(H)
    seed, offset = 0x1234567890ABCDEF, 0x1_0000_0004
    ref_in = qkv.float().requires_grad_(True)
    ref = K.attention_alibi_varlen_ref(ref_in[:, 0], ref_in[:, 1], ref_in[:, 2], cu, slopes, None, p, seed, offset, p_bf16=True)
    (ref * dout.float()).sum().backward()
    x = qkv.cuda().requires_grad_(True)
    out = ops.attn_varlen(x, cu.cuda(), max(lens), slopes.cuda(), None, p, seed, offset)
    (out.float() * dout.cuda().float()).sum().backward()
    _close(out, ref, 1e-2, 'out')                 # measured 1.9e-3 .. 2.7e-3 (round 4, oracle with flash-attn's bf16 P rounding)
    _close(x.grad, ref_in.grad, 2e-2, 'dqkv')      # measured 2.6e-3 .. 4.1e-3
    plain = ops.attn_varlen(qkv.cuda(), cu.cuda(), max(lens), slopes.cuda())
    assert (plain.float() - out.float()).abs().max().item() > 0.05          # the mask is applied


@gpu
@pytest.mark.parametrize('hd', [32, 64])
def test_attention_keep_mask_is_bit_exact(ops, hd):
    """Read the kernel's keep mask back exactly: zero scores make P uniform (1 / (i + 1)), a one-hot V window turns
    out[i, :] * (i + 1) * (1 - p) into the mask columns of that window.  Compared bit for bit with the oracle, for
    windows on and off tile boundaries and for two packed sequences (the second starts at an odd packed index)."""
    H, p = 2, 0.1
    lens = [45, 171]
    T = sum(lens)
    cu = torch.tensor([0, lens[0], T], dtype=torch.int32)
    seed, offset = 42, 20
    rate = []
    for j0 in (0, 29, 64, 139):
        q = torch.zeros(T, H, hd)
        v = torch.zeros(T, H, hd)
        for s, n in enumerate(lens):
            for j in range(j0, min(j0 + hd, n)):
                v[int(cu[s]) + j, :, j - j0] = 1.0
        qkv = torch.stack((q, q, v), dim=1).to(torch.bfloat16)
        out = ops.attn_varlen(qkv.cuda(), cu.cuda(), max(lens), None, None, p, seed, offset).float().cpu()
        for s, n in enumerate(lens):
            a = int(cu[s])
            keep = K.attn_dropout_keep(seed, offset, H, a, n, p)               # [H, n, n]
            for i in range(n):
                w = min(j0 + hd, i + 1) - j0                                   # causal part of the window
                if w <= 0:
                    assert out[a + i].abs().max() == 0
                    continue
                got = out[a + i, :, :w] * (i + 1) * (1 - p) > 0.5
                assert torch.equal(got, keep[:, i, j0:j0 + w]), (s, i, j0)
                rate.append(got.float().mean().item())
    assert abs(np.mean(rate) - 230 / 256) < 0.02


@gpu
def test_attention_backward_uses_the_forward_mask_exactly(ops):
    """dV of a one-hot dO is P^T masked: with zero scores, dV[j, d] for dO = e_d at query i equals keep(i, j) / ((i + 1)(1 - p));
    the dK/dV kernel's mask (hashes exchanged inside lane quads) is read back bit for bit."""
    H, hd, n, p = 2, 32, 100, 0.1
    seed, offset = 7, 4
    cu = torch.tensor([0, n], dtype=torch.int32)
    qkv = torch.zeros(n, 3, H, hd, dtype=torch.bfloat16)
    keep = K.attn_dropout_keep(seed, offset, H, 0, n, p)
    for i0 in (0, 31, 68):
        x = qkv.cuda().requires_grad_(True)
        out = ops.attn_varlen(x, cu.cuda(), n, None, None, p, seed, offset)
        dout = torch.zeros(n, H, hd)
        for d in range(hd):
            if i0 + d < n:
                dout[i0 + d, :, d] = 1.0
        out.backward(dout.cuda().to(torch.bfloat16))
        dv = x.grad[:, 2].float().cpu()                                          # [n, H, hd]: dv[j, h, d] = keep(i0 + d, j) / ((i0 + d + 1)(1 - p))
        for d in range(min(hd, n - i0)):
            i = i0 + d
            got = dv[: i + 1, :, d] * (i + 1) * (1 - p) > 0.5
            assert torch.equal(got.t(), keep[:, i, : i + 1]), (i0, d)
            assert dv[i + 1:, :, d].abs().sum() == 0


@gpu
@pytest.mark.parametrize('n,p', [(1 << 20, 0.1), (1003, 0.5), (7, 0.1), (4096 * 3 + 2, 0.25)])
def test_counter_dropout_kernel_equals_oracle(ops, n, p):
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, generator=g)
    seed, offset = 0xDEADBEEF12345678, 0xFFFFFFFF0 + n
    ref = K.dropout_ref(x, p, seed, offset)
    xg = x.cuda().requires_grad_(True)
    y = ops.counter_dropout(xg, p, seed, offset)
    assert torch.equal(y.detach().cpu() == 0, ref == 0)
    assert torch.allclose(y.detach().cpu(), ref, rtol=3e-7, atol=0)
    w = torch.randn(n, generator=g)
    y.backward(w.cuda())
    assert torch.allclose(xg.grad.cpu(), K.dropout_ref(w, p, seed, offset), rtol=3e-7, atol=0)
    if n >= 1 << 20:
        assert abs((y != 0).float().mean().item() - (1 - p)) < 2e-3
    assert ops.counter_dropout(xg, 0.0) is xg


@gpu
def test_dropout_counter_follows_the_torch_generator(ops):
    torch.manual_seed(31)
    dev = torch.device('cuda', 0)
    a = ops.dropout_counter(dev)
    b = ops.dropout_counter(dev)
    assert a[0] == 31 and b == (31, a[1] + 4)
    torch.manual_seed(31)
    assert ops.dropout_counter(dev) == a


def _cfg2_layer(D, B, L, table, seed):
    from offpolicy_rnn.models.rnn_base import RNNBase
    torch.manual_seed(seed)
    net = RNNBase(D, D, [], ['linear'], [CFG2])
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    return net, sd, torch.randn(B, L, D), torch.randn(B, L, D)


@gpu
def test_config2_layer_training_mode_fwd_bwd_vs_oracle():
    """The literal BASELINE configs[2] layer id at D = 256 (8 heads x 32, 6 blocks, RMSNorm, p = 0.1) in .train() mode on cuda:0
    against the oracle with the same 24 counter draws: output, dx and every parameter gradient (bf16 tolerances); .eval() is
    unchanged by the dropout code (equals the p-free oracle)."""
    from offpolicy_rnn.hip import ops
    from offpolicy_rnn.models.flash_attention.TransformerFlashAttention import PackedSeqs
    D, B, L = 256, 3, 150
    table = np.zeros((B, L), dtype=np.int64)
    table[0, :3], table[1, :2], table[2, :1] = (1, 90, 59), (70, 80), (150,)
    net, sd, x, w = _cfg2_layer(D, B, L, table, 3)
    spec = dict(layer_type=[CFG2], activation=['linear'])
    xr = x.clone().requires_grad_(True)
    pr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    torch.manual_seed(77)
    seed, off0 = ops.dropout_counter(torch.device('cuda', 0))
    ref = NW.rnn_base_forward(pr, spec, xr, NW.Flags(seqlens=torch.from_numpy(table), dropout=K.DropCounter(seed, off0 + 4)))
    (ref * w).sum().backward()
    ref_eval = NW.rnn_base_forward(sd, spec, x, NW.Flags(seqlens=torch.from_numpy(table)))
    net.to('cuda')
    net.train()
    xg = x.clone().cuda().requires_grad_(True)
    hid = net.make_init_state(B, torch.device('cuda'))
    hid.set_attention_concat_mask(PackedSeqs(table, L, torch.device('cuda')))
    y, _, _ = net.meta_forward(xg, hid)
    assert ops.dropout_counter(torch.device('cuda', 0))[1] == off0 + 4 + 4 * 24      # 6 blocks x 4 sites
    (y * w.cuda()).sum().backward()
    _close(y, ref, 1e-2, 'y')                     # north_star's bf16 bar; measured 1.3e-3 for the 6-layer configs[2] layer in training mode
    _close(xg.grad, xr.grad, 2e-2, 'dx')           # measured 4.7e-3
    for k, p in net.named_parameters():
        g_ref = pr[k].grad
        _close(p.grad, g_ref, 2e-2, 'd ' + k)       # measured <= 6.3e-3 over all 57 parameter tensors
    net.eval()
    with torch.no_grad():
        y_eval = net.meta_forward(x.cuda(), hid)[0]
    _close(y_eval, ref_eval, 1e-2, 'eval')         # measured 9.4e-4
    assert (y_eval.cpu() - y.detach().cpu()).abs().max().item() > 0.05 * ref.abs().max().item()


@gpu
def test_config2_td3_update_full_size_finite_and_first_rows_vs_oracle():
    """BASELINE configs[2] at its own size: `cgpt_h8_l6_p0.1_ml1024_rms` TD3-REDQ, 32 rows x T = 1024, D = 256, one update on
    cuda:0 - every logged scalar finite, parameters moved.  Then the value comparison the size allows: the training-mode
    layer forward of the first two rows of such a batch (T' = 1027 tokens each; packed token indices, hence masks, coincide
    with those rows inside the 32-row batch) against the oracle."""
    import math
    from bench import build_trainer
    from offpolicy_rnn.hip import ops
    from offpolicy_rnn.models.flash_attention.TransformerFlashAttention import PackedSeqs
    alg = build_trainer(CFG2, B=32, T=1024, seed=0, algo='td3')
    before = alg.policy.store.flat.clone()
    log = alg.train_one_batch()
    assert log['real_batch_size'] == 32 * 1024
    for k, v in log.items():
        v = v[0] if isinstance(v, tuple) else v
        assert math.isfinite(v), k
    assert not torch.equal(before, alg.policy.store.flat)
    del alg
    torch.cuda.empty_cache()
    # 2-row slice at the full row length, through a fresh RNNBase holding just the configs[2] layer
    D, L, B = 256, 1027, 2
    table = np.zeros((B, L), dtype=np.int64)
    table[:, 0], table[:, 1] = 1, L - 1
    net, sd, x, _ = _cfg2_layer(D, B, L, table, 5)
    x = x * 0.5
    spec = dict(layer_type=[CFG2], activation=['linear'])
    torch.manual_seed(9)
    seed, off0 = ops.dropout_counter(torch.device('cuda', 0))
    with torch.no_grad():
        ref = NW.rnn_base_forward(sd, spec, x, NW.Flags(seqlens=torch.from_numpy(table), dropout=K.DropCounter(seed, off0 + 4)))
    net.to('cuda')
    net.train()
    hid = net.make_init_state(B, torch.device('cuda'))
    hid.set_attention_concat_mask(PackedSeqs(table, L, torch.device('cuda')))
    with torch.no_grad():
        y = net.meta_forward(x.cuda(), hid)[0]
    _close(y, ref, 1e-2, 'configs[2] rows 0-1')    # measured 2.1e-3 at the full row length (T' = 1027), training mode


@pytest.mark.gpu
@pytest.mark.parametrize('p', [0.0, 0.1, 0.5])
@pytest.mark.parametrize('shape', [(257, 1024), (33, 7)])
def test_fused_gelu_dropout_equals_gelu_then_counter_dropout(p, shape):
    """`resel_gelu_dropout_fwd / _bwd` (FFN hidden of the cgpt block, reference TransformerFlashAttention.py:46-53) against torch's erf
    GELU followed by the element-wise counter dropout with the SAME (seed, offset): identical keep mask (zeros at the same elements),
    values and input gradient at 1e-6; p = 0 is plain GELU."""
    import torch
    from offpolicy_rnn.hip import ops
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(*shape, generator=g) * 2.0).cuda()
    w = torch.randn(*shape, generator=g).cuda()
    seed, offset = 1234567, 8
    xr = x.clone().requires_grad_(True)
    ref = torch.nn.functional.gelu(xr)
    ref = ops.counter_dropout(ref, p, seed, offset)
    (ref * w).sum().backward()
    xs = x.clone().requires_grad_(True)
    out = ops.gelu_dropout(xs, p, seed, offset)
    (out * w).sum().backward()
    if p > 0:
        assert torch.equal(out == 0, ref == 0) or ((out == 0) ^ (ref == 0)).sum().item() <= 2      # gelu(x) == 0 exactly is the only other zero
        keep = (ref != 0).float().mean().item()
        assert x.numel() < 100000 or abs(keep - (1 - p)) < 0.01
    scale = ref.abs().max().item()
    assert (out - ref).abs().max().item() <= 2e-6 * scale
    assert (xs.grad - xr.grad).abs().max().item() <= 2e-6 * xr.grad.abs().max().item()


@pytest.mark.gpu
def test_dropout_offset_base_shifts_every_mask_kernel():
    """`resel_dropout_offset_base`: with the device word holding k, every counter-keyed mask kernel draws the mask of offset + k -
    the element-wise dropout, the fused GELU + dropout pair and the attention forward / backward (GraphedUpdate advances the word with
    a node of its graph)."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from offpolicy_rnn.hip import ops
    dev = torch.device('cuda')
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(4099, device=dev, generator=g)
    word = torch.zeros(1, dtype=torch.int64, device=dev)
    T, H, hd = 96, 2, 32
    qkv = (torch.randn(T, 3, H, hd, device=dev, generator=g) * 0.5).to(torch.bfloat16)
    cu = torch.tensor([0, 40, 96], dtype=torch.int32, device=dev)
    slopes = torch.tensor([0.5, 0.25], device=dev)

    def run(offset):
        q = qkv.clone().requires_grad_(True)
        o = ops.attn_varlen(q, cu, 56, slopes, None, 0.3, 77, offset)
        o.float().square().sum().backward()
        xg = x.clone().requires_grad_(True)
        y = ops.gelu_dropout(xg, 0.3, 77, offset)
        y.sum().backward()
        return ops.counter_dropout(x, 0.3, 77, offset), y.detach(), xg.grad, o.detach().float(), q.grad.float()

    want = run(40)
    base_off = run(8)
    try:
        ops.dropout_offset_base(word)
        word.fill_(32)
        got = run(8)
        word.fill_(0)
        same = run(8)
    finally:
        ops.dropout_offset_base(None)
    for a, b, c, d in zip(got, want, same, base_off):
        assert torch.equal(a, b) and torch.equal(c, d)
    assert not torch.equal(got[0], base_off[0]) and not torch.equal(got[3], base_off[3])
