"""Does the row stride of the operands / the output matter to resel_gemm_f32 (L2 channel aliasing of power-of-two-ish strides)?
Same shape, mode 2 with the magnitudes given, operands and output allocated with a padded leading dimension."""
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


def padded(rows, cols, pad, scale=1.0):
    buf = torch.randn(rows, cols + pad, device=dev) * scale
    return buf[:, :cols]


print(f'{"shape":40s} ' + ' '.join(f'{h:>11s}' for h in ('ld=K,N', 'lda+32', 'lda+16', 'ldc+32', 'both+32', 'lda+4')))
for name, (M, N, K), akc, bkc in [('fwd 384 -> 2048', (T, 2048, 384), True, True), ('fwd 256 -> 1024', (T, 1024, 256), True, True),
                                  ('fwd 256 -> 256', (T, 256, 256), True, True), ('fwd 512 -> 256', (T, 256, 512), True, True),
                                  ('dgrad 1024 -> 256', (T, 256, 1024), True, False), ('dgrad 2048 -> 384', (T, 384, 2048), True, False),
                                  ('wgrad [T,1024]^T [T,256]', (1024, 256, T), False, False)]:
    row = []
    for pa, pc in ((0, 0), (32, 0), (16, 0), (0, 32), (32, 32), (4, 0)):
        A = padded(M, K, pa) if akc else padded(K, M, pa)
        B = (padded(N, K, 0, K ** -0.5) if bkc else padded(K, N, pa if not akc else 0, K ** -0.5))
        out = padded(M, N, pc)
        aa, ab = ops.amax(A.contiguous()), ops.amax(B.contiguous())
        row.append(timeit(lambda: ops.gemm_f32(A, B, akc, bkc, out=out, split=2, amax_a=aa, amax_b=ab)))
    print(f'{name:40s} ' + ' '.join(f'{t:11.1f}' for t in row))
