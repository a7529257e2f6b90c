"""Standalone driver of the var-len attention kernels at config-3 shapes (32 rows: sequences of 1 and 1026 tokens, H 8, hd 32)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'recurrent-offpolicy-rl_amd')]
import torch
from offpolicy_rnn.hip import ops
rows, H, hd = 32, 8, 32
lens = {'alt': [1, 1026] * rows, 'short_first': [1] * rows + [1026] * rows, 'long_first': [1026] * rows + [1] * rows,
        'mixed': [1 + (i * 389) % 1026 for i in range(2 * rows)]}[os.environ.get('PROF_LENS', 'alt')]
cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device='cuda')
T = sum(lens)
qkv = (torch.randn(T, 3, H, hd, device='cuda') * 0.5).to(torch.bfloat16).requires_grad_(True)
slopes = torch.tensor([2.0 ** (-8.0 * (i + 1) / H) for i in range(H)], device='cuda')
for _ in range(5):
    out = ops.attn_varlen(qkv, cu, max(lens), slopes)
    out.backward(torch.ones_like(out))
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    with torch.no_grad():
        ops.attn_varlen(qkv, cu, max(lens), slopes)
b.record(); torch.cuda.synchronize()
print(f'fwd {a.elapsed_time(b) / 20 * 1e3:.1f} us')
# per-kernel averages (HIP events around each dispatch), without and with attention-probability dropout
for p_drop in (0.0, 0.1):
    ops.profile_enable(True); ops.profile_collect()
    for _ in range(10):
        out = ops.attn_varlen(qkv, cu, max(lens), slopes, p_drop=p_drop, seed=7)
        out.backward(torch.ones_like(out))
    torch.cuda.synchronize()
    prof = ops.profile_collect(); ops.profile_enable(False)
    flops = 2.0 * 1026 ** 2 * rows * H * hd          # one product over the causal half, x2 products
    print(f'p_drop {p_drop}: ' + '  '.join(f'{k[5:-7]} {v[1]:.1f} us ({flops * m / v[1] * 1e-6:.0f} TF/s)'
                                        for (k, v), m in zip(sorted(prof.items(), key=lambda kv: ('fwd', 'dq', 'dkv').index(kv[0][5:-7])), (1, 1.5, 2))))
