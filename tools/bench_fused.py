"""The two fused-epilogue GEMMs of the critic head (resel_gemm_f32_head / resel_gemm_f32_dact) against the kernel pairs they replace,
at configs[1]'s critic shapes (8 members x 66 752 tokens x 256 x 256; activations in the shared layer's [M, E H] layout).
usage: bench_fused.py [tokens]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'recurrent-offpolicy-rl_amd')]
import torch
from offpolicy_rnn.hip import ops

dev = 'cuda'
T = int(sys.argv[1]) if len(sys.argv) > 1 else 66752
E, H = 8, 256
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g)


def timeit(fn, n=10, warm=3):
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


y2 = torch.nn.functional.elu(r(T, E * H))                       # the shared layer's output a1 as [M, E H]
a1 = y2.view(T, E, H).transpose(0, 1)                           # [E, M, H] view
W2, b2, w3, b3 = r(E, H, H) / 16, r(E, H) * 0.3, r(E, H) / 16, r(E)
h_a1, h_w2 = ops.amax(y2), ops.amax(W2)

# ---- forward: per-member GEMM + head pass  vs  resel_gemm_f32_head
def fwd_pair():
    a = ops.gemm_f32(a1, W2, True, False, amax_a=h_a1, amax_b=h_w2)
    return a, ops.ensemble_head_fwd_(a, b2, w3, b3)


def fwd_fused():
    return ops.gemm_f32_head(a1, W2, False, b2, w3, b3, amax_a=h_a1, amax_b=h_w2)


(a_p, q_p), (a_f, q_f) = fwd_pair(), fwd_fused()
print('head: max |a diff|', (a_p - a_f).abs().max().item(), 'max |q diff|', (q_p - q_f).abs().max().item(), 'scale', q_p.abs().max().item())
t_gemm = timeit(lambda: ops.gemm_f32(a1, W2, True, False, amax_a=h_a1, amax_b=h_w2))
print(f'forward : GEMM alone {t_gemm:7.1f} us | GEMM + head pass {timeit(fwd_pair):7.1f} us | fused {timeit(fwd_fused):7.1f} us')

# ---- backward: per-member input-gradient GEMM + ELU-backward / bias-gradient pass  vs  resel_gemm_f32_dact
gy = r(E, T, H)
h_gy = ops.amax(gy)


def bwd_pair():
    dx = torch.empty(T, E, H, device=dev).transpose(0, 1)
    ops.gemm_f32(gy, W2, True, True, out=dx, amax_a=h_gy, amax_b=h_w2)
    return ops.bias_act_bwd(dx.transpose(0, 1).reshape(T, E * H), y2, T, 'elu', True)


def bwd_fused():
    dx = torch.empty(T, E, H, device=dev).transpose(0, 1)
    return ops.gemm_f32_dact(gy, W2, True, a1, dx, True, amax_a=h_gy, amax_b=h_w2)


(g_p, db_p), (g_f, db_f) = bwd_pair(), bwd_fused()
g_f2 = g_f.transpose(0, 1).reshape(T, E * H)
print('dact: max |g diff|', (g_p - g_f2).abs().max().item(), 'max |db diff|', (db_p.reshape(-1) - db_f.reshape(-1)).abs().max().item(), 'scale', db_p.abs().max().item())
dx0 = torch.empty(T, E, H, device=dev).transpose(0, 1)
t_gemm = timeit(lambda: ops.gemm_f32(gy, W2, True, True, out=dx0, amax_a=h_gy, amax_b=h_w2))
print(f'backward: GEMM alone {t_gemm:7.1f} us | GEMM + ELU-backward pass {timeit(bwd_pair):7.1f} us | fused {timeit(bwd_fused):7.1f} us')
