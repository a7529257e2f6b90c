import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'recurrent-offpolicy-rl_amd')]
import torch
from offpolicy_rnn.hip import ops
B, L, Di, N = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 1043, 512, 32      # prof_sscan.py [rows]; RESEL_SSCAN_FWD_BIG=0 / 1: the TC = 16 three-workgroup forward for rows >= 96
dev = 'cuda'
xz = torch.randn(B, L, 2 * Di, device=dev).requires_grad_(True)
xdbl = torch.randn(B, L, 16 + 2 * N, device=dev).requires_grad_(True)
delta = (torch.randn(B, L, Di, device=dev) * 0.5).requires_grad_(True)
A = (-torch.exp(torch.randn(Di, N, device=dev) * 0.3)).requires_grad_(True)
Dp, db = torch.randn(Di, device=dev).requires_grad_(True), (torch.randn(Di, device=dev) * 0.1).requires_grad_(True)
start = torch.zeros(B, L, device=dev); start[:, :18] = 1
ops.profile_enable(True)
for _ in range(4):
    o = ops.selective_scan_tm(xz[..., :Di], delta, A, xdbl[..., 16:16 + N], xdbl[..., 16 + N:], Dp, xz[..., Di:], db, start, True)
    o.backward(delta.detach())
torch.cuda.synchronize()
print(ops.profile_collect())
