"""Per-kernel timings at BASELINE config-2 sizes (B=64, L=1043, Di=512, N=32).  GPU box only."""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'recurrent-offpolicy-rl_amd')]
import torch
from offpolicy_rnn.hip import ops

def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us

B, L, Di, N, D = 64, 1043, 512, 32, 256
dev = 'cuda'
xz = torch.randn(B, L, 2 * Di, device=dev)
xdbl = torch.randn(B, L, 16 + 2 * N, device=dev)
delta = torch.randn(B, L, Di, device=dev) * 0.5
A = -torch.exp(torch.randn(Di, N, device=dev) * 0.3)
Dp, db = torch.randn(Di, device=dev), torch.randn(Di, device=dev) * 0.1
start = torch.zeros(B, L, device=dev); start[:, :18] = 1
u, z = xz[..., :Di], xz[..., Di:]
Bm, Cm = xdbl[..., 16:16 + N], xdbl[..., 16 + N:]
res = {}
fwd_bytes = 4 * B * Di * L * 4 + 4 * B * N * L * 2 + B * L
bwd_bytes = 4 * B * Di * L * 7 + 4 * B * N * L * 4
t = timeit(lambda: ops.selective_scan_tm(u, delta, A, Bm, Cm, Dp, z, db, start, True))
res['sscan_fwd_us'] = t; res['sscan_fwd_GBs'] = fwd_bytes / t / 1e3
if len(sys.argv) > 1 and sys.argv[1] == 'sscan_only':
    print(json.dumps(res)); sys.exit(0)
ins = [t_.clone().requires_grad_(True) for t_ in (xz, xdbl, delta, A, Dp, db)]
def fb():
    for t_ in ins: t_.grad = None
    o = ops.selective_scan_tm(ins[0][..., :Di], ins[2], ins[3], ins[1][..., 16:16 + N], ins[1][..., 16 + N:], ins[4], ins[0][..., Di:], ins[5], start, True)
    o.backward(delta)
tfb = timeit(fb, n=10)
res['sscan_fwd+bwd_us'] = tfb
w, bconv = torch.randn(Di, 1, 16, device=dev) * 0.2, torch.randn(Di, device=dev) * 0.1
mask = torch.ones(B, L, device=dev)
t = timeit(lambda: ops.causal_conv1d_fn(u, w, bconv, mask, True))
res['conv_fwd_us'] = t; res['conv_fwd_GBs'] = 2 * 4 * B * L * Di / t / 1e3
x = torch.randn(B * L, D, device=dev); r = torch.randn(B * L, D, device=dev)
lw, lb = torch.ones(D, device=dev), torch.zeros(D, device=dev)
t = timeit(lambda: ops.layer_norm_fn(x, lw, lb, residual=r, eps=1e-8, prenorm=True))
res['addln_fwd_us'] = t; res['addln_fwd_GBs'] = 4 * 4 * B * L * D / t / 1e3
# gilr / lru at config-5 size
B5, L5, C5 = 16, 2003, 256
v, f = torch.randn(B5, L5, C5, device=dev), torch.randn(B5, L5, C5, device=dev)
st5 = torch.zeros(B5, L5, device=dev); st5[:, :2] = 1
t = timeit(lambda: ops.gilr_scan(v, f, st5, None, True))
res['gilr_fwd_us'] = t; res['gilr_fwd_GBs'] = 3 * 4 * B5 * L5 * C5 / t / 1e3
# GRU at B=64, T'=1027, H=256
gi = torch.randn(64, 1027, 768, device=dev) * 0.3
whh, bhh = torch.randn(768, 256, device=dev) / 16, torch.zeros(768, device=dev)
t = timeit(lambda: ops.gru_seq(gi, whh, bhh), n=3, warm=1)
res['gru_fwd_us'] = t; res['gru_fwd_us_per_step'] = t / 1027
print(json.dumps(res, indent=1))
