"""Kernel-alone times of the mode-2 GEMM on the tall shapes of configs[1] under the edition RESEL_GEMM_EDITION selects (read once at library
load: run once per edition) + agreement with mode 6.  usage: RESEL_GEMM_EDITION=3|4 python tools/ab_edition.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'recurrent-offpolicy-rl_amd')]
import torch
from offpolicy_rnn.hip import ops
T = 66752
g = torch.Generator(device='cuda').manual_seed(0)
r = lambda *s: torch.randn(*s, device='cuda', generator=g)


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print('edition', os.environ.get('RESEL_GEMM_EDITION', '3'))
for (N, K, bkc, act) in [(1024, 256, True, None), (2048, 384, False, 'elu'), (256, 512, True, None), (256, 1024, False, None), (256, 384, True, 'elu'),
                         (256, 256, True, 'elu'), (512, 256, False, None), (384, 2048, True, None), (512, 384, False, 'elu')]:
    A = r(T, K)
    B = (r(N, K) if bkc else r(K, N)) / K ** 0.5
    bias = r(N) if act else None
    ha, hb = ops.amax(A), ops.amax(B)
    fn = lambda: ops.gemm_f32(A, B, True, bkc, bias, act, split=2, amax_a=ha, amax_b=hb)
    ref = ops.gemm_f32(A, B, True, bkc, bias, act, split=6)
    err = ((fn() - ref).abs().max() / ref.abs().max()).item()
    print(f'[T,{K}] -> {N} b_kcontig={int(bkc)} act={act}: {timeit(fn):7.1f} us   max rel diff vs mode 6 {err:.1e}')
