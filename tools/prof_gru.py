"""The persistent GRU scans stand-alone: forward and backward time per step at B x L x H (default 64 x 1043 x 256), HIP events.  GPU box."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'recurrent-offpolicy-rl_amd')]
import torch
from offpolicy_rnn.hip import ops
B, L, H = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 1043, 256)
g = torch.Generator(device='cuda').manual_seed(0)
gi = torch.randn(B, L, 3 * H, device='cuda', generator=g).requires_grad_(True)
whh = (torch.randn(3 * H, H, device='cuda', generator=g) / H ** 0.5).requires_grad_(True)
bhh = torch.zeros(3 * H, device='cuda', requires_grad=True)
ref = torch.nn.GRU(H, H, batch_first=True).cuda()
for _ in range(2):
    out = ops.gru_seq(gi, whh, bhh)
    y = out[0] if isinstance(out, tuple) else out
    y.sum().backward()
torch.cuda.synchronize()
ops.profile_enable(True); ops.profile_collect()
for _ in range(5):
    out = ops.gru_seq(gi, whh, bhh)
    y = out[0] if isinstance(out, tuple) else out
    y.sum().backward()
torch.cuda.synchronize()
prof = ops.profile_collect(); ops.profile_enable(False)
print(f'B {B} L {L} H {H}: ' +
      '  '.join(f'{k} {v[1] / 1e3:.3f} ms = {v[1] / L:.2f} us/step' for k, v in prof.items() if k.startswith('gru')))
