"""Standalone driver of the gilr / lru scans at bench shapes (B 64, T' 1027, C 256)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'recurrent-offpolicy-rl_amd')]
import torch
from offpolicy_rnn.hip import ops
B, L, C = 64, 1027, 256
v = torch.randn(B, L, C, device='cuda', requires_grad=True)
f = torch.randn(B, L, C, device='cuda', requires_grad=True)
start = torch.zeros(B, L, device='cuda'); start[:, :2] = 1
for _ in range(5):
    h = ops.gilr_scan(v, f, start, None, True)
    h.backward(torch.ones_like(h))
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    with torch.no_grad():
        ops.gilr_scan(v, f, start, None, True)
b.record(); torch.cuda.synchronize()
t = a.elapsed_time(b) / 20 * 1e3
print(f'gilr fwd {t:.1f} us  {3 * 4 * B * L * C / t / 1e6:.2f} TB/s')
