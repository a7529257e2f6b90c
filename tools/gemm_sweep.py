"""Correctness sweep of the library's mode-2 GEMM against mode 6 on grids from one block to more tiles than CUs, all four operand
layouts, batched (ensemble) and K-split (weight-gradient) shapes: the unit tests' sizes do not fill the chip, and a pipeline that reads a
register before its load has landed only fails on large grids.  RESEL_GEMM_EDITION selects the edition under test."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'recurrent-offpolicy-rl_amd')]
import torch
from offpolicy_rnn.hip import ops
g = torch.Generator(device='cuda').manual_seed(0)
bad = n = 0


def check(tag, A, B, akc, bkc, bias=None, act=None, out0=None):
    global bad, n
    aa, ab = ops.amax(A.reshape(-1, A.shape[-1])), ops.amax(B.reshape(-1, B.shape[-1]))
    out = ops.gemm_f32(A, B, akc, bkc, bias, act, out=None if out0 is None else out0.clone(), split=2, amax_a=aa, amax_b=ab)
    ref = ops.gemm_f32(A, B, akc, bkc, bias, act, out=None if out0 is None else out0.clone(), split=6)
    err = (out - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)
    n += 1
    if not err < 1e-5:
        bad += 1
        print('BAD', tag, tuple(A.shape), tuple(B.shape), akc, bkc, err, flush=True)


r = lambda *s: torch.randn(*s, device='cuda', generator=g)
for M in (256, 4096, 66752):
    for N in (128, 256, 1024):
        for K in (64, 256, 384):
            check('fwd', r(M, K), r(N, K) / K ** 0.5, True, True, r(N), 'elu')
            check('dgrad', r(M, K), r(K, N) / K ** 0.5, True, False)
for (Mo, No) in ((256, 384), (1024, 256), (2048, 384), (128, 256)):
    for T in (4096, 66752):
        check('wgrad', r(T, Mo), r(T, No), False, False)
        check('a-transposed', r(T, Mo), r(No, T) / T ** 0.5, False, True)
for M in (32800, 32016, 1000):                          # row counts that are no multiple of 64 (cgpt's shifted pass, gilr / lru at T = 2000)
    for N in (80, 256, 200):
        check('ragged fwd', r(M, 256), r(N, 256) / 16, True, True, r(N))
        check('ragged dgrad', r(M, 256), r(256, N) / 16, True, False)
        check('ragged wgrad', r(M, 256), r(M, N), False, False)
check('accumulate', r(66752, 128), r(128, 512) / 11, True, False, None, ops.GEMM_ACCUMULATE, r(66752, 512))
check('accumulate ragged', r(32800, 256), r(200, 256) / 16, True, True, None, ops.GEMM_ACCUMULATE, r(32800, 200))
check('softplus', r(66752, 64), r(512, 64) / 8, True, True, r(512), ops.GEMM_SOFTPLUS)
check('efc-8 fwd', r(8, 66752, 256), r(8, 256, 256) / 16, True, False, r(8, 256), 'elu')
check('efc-8 fwd kc', r(8, 66752, 256), r(8, 256, 256) / 16, True, True, r(8, 256))
check('efc-8 wgrad', r(8, 66752, 256), r(8, 66752, 256), False, False)
print('sweep:', f'{n} cases ok' if not bad else f'{bad} of {n} cases wrong')
sys.exit(1 if bad else 0)
