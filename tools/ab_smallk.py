"""Small-K accumulate GEMM (x_proj input gradient: [T, 80] x [80, 512] added into dxc): product modes / kernel editions, us per call."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'recurrent-offpolicy-rl_amd')]
import torch
from offpolicy_rnn.hip import ops
T = 66752
g = torch.Generator(device='cuda').manual_seed(0)
r = lambda *s: torch.randn(*s, device='cuda', generator=g)


def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (M, N, K, akc, bkc, act) in [(T, 512, 80, True, False, ops.GEMM_ACCUMULATE), (T, 512, 80, True, False, None), (T, 384, 40, True, True, None),
                                 (T, 16, 512, True, False, None), (T, 128, 256, True, True, None), (T, 512, 16, True, True, None)]:
    A = r(M, K)
    B = r(N, K) if bkc else r(K, N)
    out = r(M, N)
    res = []
    for sp in (0, 6, 106, 3):
        try:
            res.append((sp, timeit(lambda: ops.gemm_f32(A, B, akc, bkc, None, act, out=out, split=sp))))
        except Exception as e:
            res.append((sp, str(e)[:30]))
    lib = timeit((lambda: out.addmm_(A, B if not bkc else B.t())) if act else (lambda: torch.mm(A, B if not bkc else B.t(), out=out)))
    print(f'M{M} N{N} K{K} a{int(akc)} b{int(bkc)} {act}: ' + '  '.join(f'mode {sp}: {t if isinstance(t, str) else round(t, 1)}' for sp, t in res) + f'  library {lib:.1f}')
