"""A/B of resel_gemm_f32 product modes on the update's shapes (configs[1], 66 752 tokens): mode 6 (three bf16 planes, six products)
against mode 2 (two fp16 planes of the scaled operands, three products) with the operand magnitudes given (kernel alone) and
with the resel_amax pre-pass inside the call.  Error: max |C - C64| / sum|a b| on the first 1 024 rows."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'recurrent-offpolicy-rl_amd')]
import torch
from offpolicy_rnn.hip import ops
dev = 'cuda'
T = int(sys.argv[1]) if len(sys.argv) > 1 else 66752


def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g)
print(f'{"shape":46s} {"mode 6":>9s} {"mode 2":>9s} {"+ amax":>9s} {"mode 3":>9s} | err6 err2 (x 1e-8)')
cases = []
for (K, N) in [(384, 256), (256, 256), (256, 1024), (512, 256), (384, 2048)]:
    cases.append((f'fwd   [{T},{K}] -> {N}', r(T, K), r(N, K) / K ** 0.5, True, True))
for (N, K) in [(256, 384), (1024, 256), (2048, 384)]:
    cases.append((f'dgrad [{T},{N}] x [{N},{K}]', r(T, N), r(N, K) / K ** 0.5, True, False))
for (N, K) in [(256, 384), (1024, 256), (2048, 384)]:
    cases.append((f'wgrad [{T},{N}]^T x [{T},{K}]', r(T, N), r(T, K), False, False))
cases.append(('efc-8 fwd 8 x [T,256] -> 256', r(8, T, 256), r(8, 256, 256) / 16, True, False))
cases.append(('efc-8 wgrad', r(8, T, 256), r(8, T, 256), False, False))
for name, A, B, akc, bkc in cases:
    aa, ab = ops.amax(A), ops.amax(B)
    t6 = timeit(lambda: ops.gemm_f32(A, B, akc, bkc, split=6))
    t2 = timeit(lambda: ops.gemm_f32(A, B, akc, bkc, split=2, amax_a=aa, amax_b=ab))
    t2p = timeit(lambda: ops.gemm_f32(A, B, akc, bkc, split=2))
    t3 = timeit(lambda: ops.gemm_f32(A, B, akc, bkc, split=3))
    Ad = (A if akc else A.transpose(-1, -2))[..., :1024, :].double()
    Bd = (B.transpose(-1, -2) if bkc else B).double()
    ref, sc = Ad @ Bd, Ad.abs() @ Bd.abs()
    e = {}
    for sp in (6, 2):
        out = ops.gemm_f32(A, B, akc, bkc, split=sp)
        e[sp] = ((out[..., :1024, :].double() - ref).abs() / sc).max().item()
    print(f'{name:46s} {t6:9.1f} {t2:9.1f} {t2p:9.1f} {t3:9.1f} | {e[6] * 1e8:6.2f} {e[2] * 1e8:6.2f}')
