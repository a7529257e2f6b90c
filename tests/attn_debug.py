"""Error pattern of the attention kernels against the oracle on small cases (debug aid, not collected by pytest; lives under tests/ because
only tests may import oracle/).  GPU box: python3 tests/attn_debug.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'recurrent-offpolicy-rl_amd')]
import numpy as np, torch
from offpolicy_rnn.hip import ops
from oracle import kernels as K
torch.set_printoptions(linewidth=250, precision=3, sci_mode=False)
for H, hd, lens in [(1, 32, [5]), (1, 32, [40]), (1, 32, [130]), (2, 64, [70]), (8, 32, [1, 130, 37, 64])]:
    g = torch.Generator().manual_seed(1)
    T = sum(lens)
    qkv = (torch.randn(T, 3, H, hd, generator=g) * 0.8).to(torch.bfloat16)
    dout = torch.randn(T, H, hd, generator=g).to(torch.bfloat16)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
    slopes = K.alibi_slopes(H)
    ref_in = qkv.float().requires_grad_(True)
    ref = K.attention_alibi_varlen_ref(ref_in[:, 0], ref_in[:, 1], ref_in[:, 2], cu, slopes)
    (ref * dout.float()).sum().backward()
    x = qkv.cuda().requires_grad_(True)
    out = ops.attn_varlen(x, cu.cuda(), max(lens), slopes.cuda())
    (out.float() * dout.cuda().float()).sum().backward()
    e = (out.float().cpu() - ref).abs()
    print(f'H {H} hd {hd} lens {lens}: out err max {e.max():.4f} (ref max {ref.abs().max():.3f}); per-token max:', e.amax(dim=(1, 2))[:48])
    if e.max() > 0.05:
        tok = int(e.amax(dim=(1, 2)).argmax())
        print('  token', tok, 'head 0 got', out[tok, 0, :16].float().cpu(), '\n  ref', ref[tok, 0, :16].detach())
    ge = (x.grad.float().cpu() - ref_in.grad).abs()
    print('   dq err', ge[:, 0].max().item(), 'dk err', ge[:, 1].max().item(), 'dv err', ge[:, 2].max().item(), ' grad scale', ref_in.grad.abs().max().item())
