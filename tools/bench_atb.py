import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'recurrent-offpolicy-rl_amd')]
import torch
from offpolicy_rnn.hip import ops
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - a) / n * 1e6
K = 66752
for Wd, Nd, ldn, tr in [(512, 16, 80, False), (512, 80, 80, True), (256, 80, 80, True), (1024, 16, 16, False)]:
    w = torch.randn(K, Wd, device='cuda'); nf = torch.randn(K, ldn, device='cuda'); n = nf[:, :Nd]
    lib_us = t(lambda: torch.mm(n.t(), w) if tr else torch.mm(w.t(), n))
    my_us = t(lambda: ops.atb(w, n, tr))
    print(f'Wd={Wd} Nd={Nd} tr={tr}: library {lib_us:.1f} us, atb {my_us:.1f} us')
