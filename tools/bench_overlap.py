"""Does a VALU-bound scan overlap with an MFMA-bound library GEMM when both run on separate streams?"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'recurrent-offpolicy-rl_amd')]
import torch
from offpolicy_rnn.hip import ops
B, L, Di, N = 64, 1043, 512, 32
dev = 'cuda'
u = torch.randn(B, L, Di, device=dev); dt = torch.randn(B, L, Di, device=dev) * 0.1; z = torch.randn(B, L, Di, device=dev)
A = -torch.rand(Di, N, device=dev); Bm = torch.randn(B, L, N, device=dev); Cm = torch.randn(B, L, N, device=dev)
D = torch.ones(Di, device=dev); db = torch.zeros(Di, device=dev)
x = torch.randn(B * L, 256, device=dev); w = torch.randn(1024, 256, device=dev)
x2 = torch.randn(B * L, 384, device=dev); w2 = torch.randn(2048, 384, device=dev)
def scan(): return ops.selective_scan_tm(u, dt, A, Bm, Cm, D, z, db, None, True)
def gemm(): return torch.mm(x, w.t())
def gemm2(): return torch.mm(x2, w2.t())
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - a) / n * 1e6
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def both(g, k=1):
    def f():
        with torch.cuda.stream(s1):
            for _ in range(k): scan()
        with torch.cuda.stream(s2):
            g()
    return f
with torch.no_grad():
    ts, tg, tg2 = t(scan), t(gemm), t(gemm2)
    print(f'scan {ts:.0f} us, gemm[66752x256x1024] {tg:.0f} us, gemm[66752x384x2048] {tg2:.0f} us')
    print(f'scan || gemm  : {t(both(gemm)):.0f} us (sum {ts + tg:.0f}, max {max(ts, tg):.0f})')
    print(f'3 scans || gemm2: {t(both(gemm2, 3)):.0f} us (sum {3 * ts + tg2:.0f}, max {max(3 * ts, tg2):.0f})')
