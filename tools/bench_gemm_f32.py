"""resel_gemm_f32 vs the library GEMM (torch.mm / addmm / bmm -> rocBLAS / hipBLASLt, tuned table on) on the shapes of the update
at BASELINE configs[1] (66 752 tokens).  Prints time, TFLOP/s and max relative error vs an fp64 product for each shape."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'recurrent-offpolicy-rl_amd')]
import torch
from offpolicy_rnn.hip import ops
dev = 'cuda'
T = int(sys.argv[1]) if len(sys.argv) > 1 else 66752
if len(sys.argv) > 2:
    ops.GEMM_SPLIT = int(sys.argv[2])
print('tokens', T, 'product mode (split)', ops.GEMM_SPLIT)


def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def report(name, flops, t_lib, t_mine, err):
    print(f'{name:58s} lib {t_lib:8.1f} us {flops / t_lib / 1e6:6.1f} TF | mine {t_mine:8.1f} us {flops / t_mine / 1e6:6.1f} TF | x{t_lib / t_mine:5.2f} | rel err {err:.1e}')


g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g)
# forward y = act(x W^T + b)
for (K, N, act) in [(384, 256, 'elu'), (256, 256, 'elu'), (256, 128, None), (256, 1024, None), (512, 256, None), (384, 2048, 'elu'), (512, 80, None)]:
    x, W, b = r(T, K), r(N, K) / K ** 0.5, r(N)
    lib = lambda: torch.nn.functional.elu(torch.addmm(b, x, W.t())) if act else torch.addmm(b, x, W.t())
    mine = lambda: ops.gemm_f32(x, W, True, True, b, act)
    ref = (x[:2048].double() @ W.double().t() + b.double())
    ref = torch.nn.functional.elu(ref) if act else ref
    err = ((mine()[:2048].double() - ref).abs().max() / ref.abs().max()).item()
    report(f'fwd   [{T},{K}] x [{N},{K}]^T +b {act or ""}', 2.0 * T * K * N, timeit(lib), timeit(mine), err)
# dgrad dx = dy W
for (N, K) in [(256, 384), (256, 256), (1024, 256), (256, 512), (2048, 384), (128, 256)]:
    dy, W = r(T, N), r(N, K) / K ** 0.5
    lib = lambda: dy @ W
    mine = lambda: ops.gemm_f32(dy, W, True, False)
    ref = dy[:2048].double() @ W.double()
    err = ((mine()[:2048].double() - ref).abs().max() / ref.abs().max()).item()
    report(f'dgrad [{T},{N}] x [{N},{K}]', 2.0 * T * K * N, timeit(lib), timeit(mine), err)
# wgrad dW = dy^T x
for (N, K) in [(256, 384), (256, 256), (128, 256), (1024, 256), (256, 512), (2048, 384), (80, 512)]:
    dy, x = r(T, N), r(T, K)
    lib = lambda: dy.t() @ x
    mine = lambda: ops.gemm_f32(dy, x, False, False)
    ref = dy.double().t() @ x.double()
    err = ((mine().double() - ref).abs().max() / ref.abs().max()).item()
    report(f'wgrad [{T},{N}]^T x [{T},{K}]', 2.0 * T * K * N, timeit(lib), timeit(mine), err)
# ensemble layer (per member): y[e] = act(x[e] W[e] + b[e]),  W stored [E, in, out] (reference EnsembleLinear)
E = 8
x3, W3, b3 = r(E, T, 256), r(E, 256, 256) / 16, r(E, 1, 256)
lib = lambda: torch.nn.functional.elu(torch.baddbmm(b3, x3, W3))
mine = lambda: ops.gemm_f32(x3, W3, True, False, b3, 'elu')
ref = torch.nn.functional.elu(torch.baddbmm(b3.double(), x3[:, :512].double(), W3.double()))
err = ((mine()[:, :512].double() - ref).abs().max() / ref.abs().max()).item()
report(f'efc-8 [{E},{T},256] x [8,256,256] +b elu', 2.0 * E * T * 256 * 256, timeit(lib), timeit(mine), err)
g3 = r(E, T, 256)
lib = lambda: torch.bmm(g3, W3.transpose(1, 2))
mine = lambda: ops.gemm_f32(g3, W3, True, True)
ref = torch.bmm(g3[:, :512].double(), W3.double().transpose(1, 2))
err = ((mine()[:, :512].double() - ref).abs().max() / ref.abs().max()).item()
report(f'efc-8 dgrad [{E},{T},256] x [8,256,256]^T', 2.0 * E * T * 256 * 256, timeit(lib), timeit(mine), err)
lib = lambda: torch.bmm(x3.transpose(1, 2), g3)
mine = lambda: ops.gemm_f32(x3, g3, False, False)
ref = torch.bmm(x3.double().transpose(1, 2), g3.double())
err = ((mine().double() - ref).abs().max() / ref.abs().max()).item()
report(f'efc-8 wgrad [{E},{T},256]^T x [8,{T},256]', 2.0 * E * T * 256 * 256, timeit(lib), timeit(mine), err)
lib = lambda: torch.bmm(x3, W3)
mine = lambda: ops.gemm_f32(x3, W3, True, False)
report(f'efc-8 plain fwd (no epilogue)', 2.0 * E * T * 256 * 256, timeit(lib), timeit(mine), 0.0)
