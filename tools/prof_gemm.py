"""One shape of resel_gemm_f32, many repetitions (PMC / stats driver).  python tools/prof_gemm.py M N K [akc bkc] [reps]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'recurrent-offpolicy-rl_amd')]
import torch
from offpolicy_rnn.hip import ops
M, N, K = (int(a) for a in sys.argv[1:4])
akc, bkc = (bool(int(sys.argv[4])), bool(int(sys.argv[5]))) if len(sys.argv) > 5 else (True, True)
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 20
if len(sys.argv) > 7:
    ops.GEMM_SPLIT = int(sys.argv[7])
A = torch.randn((M, K) if akc else (K, M), device='cuda')
B = torch.randn((N, K) if bkc else (K, N), device='cuda')
for _ in range(reps):
    C = ops.gemm_f32(A, B, akc, bkc)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    C = ops.gemm_f32(A, B, akc, bkc)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
print(f'gemm_f32 {M}x{N}x{K}: {us:.1f} us, {2.0 * M * N * K / us / 1e6:.1f} TFLOP/s')

if os.environ.get('RESEL_GEMM_STAMPS'):
    import ctypes, numpy as np
    from offpolicy_rnn.hip._lib import lib
    buf = np.zeros(64 * 16, dtype=np.uint64)
    rc = lib().resel_gemm_debug_stamps(ctypes.c_void_p(buf.ctypes.data))
    st = buf.reshape(64, 16)[:, :9].astype(np.int64)
    names = ['q0', 'ds_write', 'q1', 'produce', 'barrier', 'frag reads', 'q2', 'q3', '-> next step']
    print('rc', rc, 'phases:', names)
    for i in range(2, 26):
        d = np.diff(st[i]).tolist() + [int(st[i + 1][0] - st[i][8])]
        print(f'step {i:2d}: ' + ' '.join(f'{x:5d}' for x in d) + f' | total {int(st[i + 1][0] - st[i][0])}')

if os.environ.get('RESEL_BF3_CLOCK'):            # tools/bf3_ablate.sh CLOCK build: in-kernel clock of the last launch
    import ctypes, numpy as np
    from offpolicy_rnn.hip._lib import lib
    buf = np.zeros(512, dtype=np.uint64)
    rc = ctypes.CDLL(os.environ['RESEL_HIP_LIBRARY']).resel_bf3_debug_clock(ctypes.c_void_p(buf.ctypes.data))
    t, r = buf[0::2].astype(np.float64), buf[1::2].astype(np.float64)
    ok = r > 0
    clk = t[ok] / r[ok] * 0.1
    print(f'in-kernel clock (s_memtime / s_memrealtime, {ok.sum()} blocks): median {np.median(clk):.3f} GHz, min {clk.min():.3f}, max {clk.max():.3f}; '
          f'block lifetime median {np.median(r[ok]) * 0.01:.1f} us')
