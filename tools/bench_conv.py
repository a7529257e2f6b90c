"""conv1d fwd/bwd timing: strided x-half of xz [M, 2Di] against a contiguous [M, Di] input (20 launches per event pair)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'recurrent-offpolicy-rl_amd')]
import torch
from offpolicy_rnn.hip import ops
B, L, Di, K = 64, 1043, 512, 16
dev = 'cuda'
torch.manual_seed(0)
xz = torch.randn(B, L, 2 * Di, device=dev)
xc = torch.randn(B, L, Di, device=dev)
w = torch.randn(Di, 1, K, device=dev) * 0.2
bias = torch.randn(Di, device=dev) * 0.1
mask = torch.ones(B, L, 1, device=dev)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, x in (('strided', xz[..., :Di]), ('contiguous', xc)):
    xr = x.detach().requires_grad_(True) if x.is_contiguous() else x
    ts = [timeit(lambda: ops.causal_conv1d_fn(x, w, bias, mask, True), 100) for _ in range(3)]
    print(f'fwd {name:10s} ' + ' '.join(f'{t:8.1f}' for t in ts) + f' us  {2 * B * L * Di * 4 / min(ts) / 1e6:6.2f} TB/s')
xg = xc.clone().requires_grad_(True)
wg = w.clone().requires_grad_(True)
y = ops.causal_conv1d_fn(xg, wg, bias, mask, True)
g = torch.randn_like(y)
ts = [timeit(lambda: torch.autograd.grad(y, (xg, wg), g, retain_graph=True), 100) for _ in range(3)]
print('bwd contiguous ' + ' '.join(f'{t:8.1f}' for t in ts) + ' us (100 launches each)')
torch.manual_seed(1)
g = torch.randn_like(y)
dx, dw = torch.autograd.grad(y, (xg, wg), g, retain_graph=True)
print(f'check: |dx| {dx.double().abs().sum().item():.9e}  dx[3,700,5] {dx[3, 700, 5].item():.9e}  |dw| {dw.double().abs().sum().item():.9e}')
