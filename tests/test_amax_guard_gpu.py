"""Guards around GEMM product mode 2 (fp16 planes of the SCALED operands, csrc/gemm_bf3.hip `f16_scale`): the scale comes from a
magnitude handle; a handle below the operand's true maximum overflows fp16 to inf.  Round 5:
  * `resel_amax` keeps no state: two pre-passes in flight on different streams cannot disturb each other;
  * RESEL_AMAX_VERIFY=1 (`ops.AMAX_VERIFY`) checks every handle in front of every mode-2 product and the trainers raise
    `AmaxBoundError` at the end of the update - one whole update per sequence layer + a replayed one must come out clean;
  * a handle deliberately 4x too small is reported (never a silent inf);
  * handles kept on `ctx` die with their arena slot; the epoch counter starts over at update boundaries."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def ops():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from offpolicy_rnn.hip import ops as o
    return o


def test_amax_prepasses_on_two_streams_do_not_mix(ops):
    """the first version shared one ticket / partial buffer per device."""
    dev = torch.device('cuda')
    g = torch.Generator(device='cpu').manual_seed(0)
    xs = [torch.randn(16384, 512, generator=g).mul_(s).to(dev) for s in (1.0, 37.0, 0.01, 1234.0)]
    want = [float(x.abs().max()) for x in xs]
    streams = [torch.cuda.Stream() for _ in xs]
    torch.cuda.synchronize()
    for rep in range(20):
        hs = []
        for x, st in zip(xs, streams):
            with torch.cuda.stream(st):
                hs.append(ops.amax(x))
        torch.cuda.synchronize()
        got = [ops.amax_value(h) for h in hs]
        assert got == want, (rep, got, want)


def test_verify_mode_reports_a_handle_that_is_4x_too_small(ops, monkeypatch):
    """(c) of VERDICT r04 item 4: the product is never handed on silently - the check in front of the mode-2 launch reports it."""
    monkeypatch.setattr(ops, 'AMAX_VERIFY', True)
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(1)
    A = torch.randn(8192, 256, generator=g).to(dev)
    W = torch.randn(384, 256, generator=g).to(dev)
    good_a, good_w = ops.amax(A), ops.amax(W)
    ops.gemm_f32(A, W, True, True, split=2, amax_a=good_a, amax_b=good_w)
    ops.amax_verify_raise(dev)                                   # true bounds: clean
    low = ops.amax(A * 0.25)                                     # a handle 4x below max |A|
    C = ops.gemm_f32(A, W, True, True, split=2, amax_a=low, amax_b=good_w)
    with pytest.raises(ops.AmaxBoundError, match='BELOW its operand'):
        ops.amax_verify_raise(dev)
    del C
    ops.amax_verify_raise(dev)                                   # the report was consumed
    # the same for the B operand, and for one member of a batched operand
    lowb = ops.amax(W * 0.25)
    ops.gemm_f32(A, W, True, True, split=2, amax_a=good_a, amax_b=lowb)
    with pytest.raises(ops.AmaxBoundError):
        ops.amax_verify_raise(dev)
    A3 = torch.randn(4, 4096, 256, generator=g).to(dev)
    W3 = torch.randn(4, 256, 256, generator=g).to(dev)
    h3 = ops.amax(A3)
    A3[2, 17, 5] = 9.0 * float(A3.abs().max())                   # an in-place write torch sees would void a TAG; an explicit handle is the caller's promise
    ops.gemm_f32(A3, W3, True, False, split=2, amax_a=h3, amax_b=ops.amax(W3))
    with pytest.raises(ops.AmaxBoundError):
        ops.amax_verify_raise(dev)


def test_kept_handles_die_with_their_arena_slot(ops):
    """a handle saved on ctx in a forward must not be used by the backward once the ring has given its slot away."""
    x = torch.randn(4096, 64, device='cuda')
    dev = x.device                                              # arenas are keyed by the tensors' own device object (cuda:0)
    h = ops.amax(x)
    ops.tag_amax(x, h)
    kept = ops.keep_handles(h, None)
    assert ops.handle_alive(kept[0]) is h and ops.handle_alive(kept[1]) is None and ops.amax_of(x) is h
    for _ in range(ops.AMAX_SLOTS - 1):
        ops.amax_slot(dev)
    assert ops.handle_alive(kept[0]) is h                        # one short of a lap: still the same tenant
    ops.amax_slot(dev)                                           # the lap completes: the slot has a new tenant
    assert ops.handle_alive(kept[0]) is None and ops.amax_of(x) is None


def test_epoch_rollover_at_an_update_boundary(ops):
    """when the 31-bit epochs start over, tags, kept handles, weight handles and captured graphs all go together;
    products after the reset are right (a stale large epoch would outrank every new publication)."""
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(2)
    lin = torch.nn.Linear(256, 384).to(dev)
    from offpolicy_rnn.models.flat_params import FlatParameterStore
    from collections import OrderedDict
    store = FlatParameterStore(OrderedDict(lin=lin))
    x = torch.randn(8192, 256, generator=g).to(dev)
    ref = (x.double() @ lin.weight.double().t() + lin.bias.double()).float()
    assert not ops.amax_maintenance()                           # far from the limit: nothing happens
    ops._AMAX_EPOCH[0] = ops._EPOCH_RESET_AT - 40               # as after ~3 million eager updates: the next products carry epochs near 2^31
    y0 = ops.linear_act(x, lin.weight, lin.bias, None)
    assert store._amax is not None and ops._AMAX_EPOCH[0] < ops._EPOCH_RESET_AT
    ops._AMAX_EPOCH[0] = ops._EPOCH_RESET_AT + 5                # past the limit
    gen = ops.AMAX_GENERATION[0]
    assert ops.amax_maintenance()                               # the update boundary resets
    assert ops.AMAX_GENERATION[0] == gen + 1 and store._amax is None and ops._AMAX_EPOCH[0] == 0
    # a long-lived user that never reaches an update boundary (evaluation loops, soak tools) is reset by the next
    # producer call itself instead of failing when the 31-bit word runs out
    ops._AMAX_EPOCH[0] = ops._EPOCH_RESET_AT + 5
    kept = ops.keep_handles(ops.amax_slot(dev)[0])              # this very call starts the epochs over ...
    assert ops.AMAX_GENERATION[0] == gen + 2 and ops._AMAX_EPOCH[0] == 1 and ops.handle_alive(kept[0]) is not None
    with torch.no_grad():
        lin.weight.mul_(64.0)                                    # the old handle (large epoch) would now be 64x too small
    ref2 = (x.double() @ lin.weight.double().t() + lin.bias.double()).float()
    y1 = ops.linear_act(x, lin.weight, lin.bias, None)
    assert torch.isfinite(y1).all()
    assert (y0 - ref).abs().max() <= 1e-5 * ref.abs().max() and (y1 - ref2).abs().max() <= 1e-5 * ref2.abs().max()


@pytest.mark.parametrize('rnn,algo', [('smamba_s32_c16_b2_nln', 'sac'), ('gilr', 'sac'), ('lru', 'sac'), ('gru', 'sac'),
                                      ('cgpt_h8_l2_p0.1_ml1024_rms', 'td3')])
def test_whole_update_in_verify_mode_is_clean(ops, monkeypatch, rnn, algo):
    """(a) + (b): every mode-2 product of two whole updates at the BASELINE width (D = 256, T = 1024, 8 rows = 8k tokens: the hand-written
    GEMMs and mode 2 are what runs) has handles that bound its operands - published by producers, weight stores and pre-passes alike."""
    from bench import build_trainer
    monkeypatch.setattr(ops, 'AMAX_VERIFY', True)
    assert ops.gemm_split() == 2
    alg = build_trainer(rnn, B=8, T=1024, seed=0, algo=algo)
    n0 = ops._VERIFY_TAG[0]
    for _ in range(2):
        log = alg.train_one_batch()                              # raises AmaxBoundError itself at the end of the update
        alg.grad_num += 1
    ops.amax_verify_raise(alg.device)
    assert ops._VERIFY_TAG[0] - n0 >= 40, 'mode 2 did not run: nothing was verified'
    for k, v in dict(log).items():
        v = v[0] if isinstance(v, tuple) else v
        assert np.isfinite(v), k


def test_replayed_update_in_verify_mode_is_clean(ops, monkeypatch):
    """(b): a GraphedUpdate replay - its kernel nodes carry the handles and epochs of the recording, the check kernels are nodes too."""
    from bench import build_trainer
    from offpolicy_rnn.algorithm.graphed_update import GraphedUpdate
    monkeypatch.setattr(ops, 'AMAX_VERIFY', True)
    alg = build_trainer('smamba_s32_c16_b2_nln', B=8, T=1024, seed=0)
    why = GraphedUpdate.refusal(alg)
    if why:
        pytest.skip(why)
    gu = GraphedUpdate(alg, warmup=1)
    try:
        for _ in range(5):
            log = gu.step()
            alg.grad_num += 1
        assert len(gu.graphs) == 1 and gu.eager_fallbacks == 1    # the warm-up update is the shape's first visit: recorded on the second, replayed four times
        for k, v in dict(log).items():
            v = v[0] if isinstance(v, tuple) else v
            assert np.isfinite(v), k
    finally:
        gu.close()


def test_more_recurring_shapes_than_graphs_do_not_thrash(ops, monkeypatch):
    """with more recurring batch shapes than `max_graphs` the overflow runs eagerly - an evicted shape is not
    recorded again on its next visit, and recordings are rate-limited."""
    from test_host_logic import _push, _synth, make_parameter
    from offpolicy_rnn import alg_init
    from offpolicy_rnn.algorithm.graphed_update import GraphedUpdate
    torch.manual_seed(0)
    np.random.seed(0)
    alg = alg_init(make_parameter('gilr', sac_batch_size=4 * 12 - 1, cuda_inference=True))
    rs = np.random.RandomState(3)
    for n in (12, 9, 7, 12, 5, 12, 10, 8, 11, 6):
        o, a, r = _synth(rs, n, 5, 3)
        _push(alg.replay_buffer, o, a, r, early_done=(n != 12))
    gu = GraphedUpdate(alg, warmup=1, max_graphs=1)
    records = 0
    real = torch.cuda.CUDAGraph

    class Counting(real):
        def __new__(cls, *a, **k):
            nonlocal records
            records += 1
            return real.__new__(cls, *a, **k)
    monkeypatch.setattr(torch.cuda, 'CUDAGraph', Counting)
    n_upd = 60
    for _ in range(n_upd):
        gu.step()
        alg.grad_num += 1
    torch.cuda.synchronize()
    shapes = sum(1 for v in gu._seen.values() if v != 0)
    print(f'{records} recordings, {gu.eager_fallbacks} eager of {n_upd}, {shapes} shapes seen')
    assert len(gu.graphs) <= 1
    assert records <= 1 + n_upd // GraphedUpdate.RECAPTURE_HITS, 'an evicted shape was recorded again right away'
    assert len(gu._seen) <= GraphedUpdate.SEEN_CAP
