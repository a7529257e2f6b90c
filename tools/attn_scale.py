"""Forward attention time vs the number of sequences (critical path of one workgroup vs throughput).  GPU box."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'recurrent-offpolicy-rl_amd')]
import torch
from offpolicy_rnn.hip import ops
H, hd = 8, 32
L = int(sys.argv[1]) if len(sys.argv) > 1 else 1026
slopes = torch.tensor([2.0 ** (-8.0 * (i + 1) / H) for i in range(H)], device='cuda')
for S in ([int(a) for a in sys.argv[2].split(',')] if len(sys.argv) > 2 else (1, 4, 16, 32, 64, 128)):
    lens = [L] * S
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device='cuda')
    qkv = (torch.randn(sum(lens), 3, H, hd, device='cuda') * 0.5).to(torch.bfloat16).requires_grad_(True)
    res = {}
    ops.profile_enable(True); ops.profile_collect()
    for _ in range(12):
        out = ops.attn_varlen(qkv, cu, L, slopes)
        out.backward(torch.ones_like(out))
    torch.cuda.synchronize()
    prof = ops.profile_collect(); ops.profile_enable(False)
    fl = 2.0 * L * L * S * H * hd
    print(f'S {S:4d} len {L}: ' + '  '.join(f'{k[5:-7]} {v[1]:7.1f} us' for k, v in prof.items()) + f'   fwd {fl / prof["attn_fwd_kernel"][1] * 1e-6:.0f} TF/s')
