"""Full-size A/B of the GEMM product modes: same seeds, N updates of the bench configuration with RESEL_GEMM_SPLIT = 6 / 9 / 0 and with the
library GEMMs (RESEL_GEMM_F32_MIN_ROWS huge); logged scalars and a parameter checksum side by side.  The reference point is the
library run: how far is each mode from it, and how far are two library-free fp32 modes from each other (accumulation order noise)."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = int(sys.argv[2]) if len(sys.argv) > 2 else 5
code = r'''
import sys, os, json, numpy as np, torch
sys.path[:0] = [%r, os.path.join(%r, 'recurrent-offpolicy-rl_amd')]
from bench import build_trainer
torch.manual_seed(1); np.random.seed(1)
alg = build_trainer(sys.argv[1], 64, 1024)
torch.manual_seed(2); np.random.seed(2)
out = []
for _ in range(int(sys.argv[2])):
    log = dict(alg.train_one_batch()); alg.grad_num += 1
    out.append({k: float(v[0] if isinstance(v, tuple) else v) for k, v in log.items()})
print(json.dumps(dict(logs=out, psum=float(alg.policy.store.flat.double().abs().sum()), vsum=float(alg.values[0].store.flat.double().abs().sum()))))
''' % (ROOT, ROOT)
rnn = sys.argv[1] if len(sys.argv) > 1 else 'smamba_s32_c16_b2_nln'
runs = {'library': dict(RESEL_GEMM_F32_MIN_ROWS='1000000000'), 'split 6': dict(RESEL_GEMM_SPLIT='6'), 'split 9': dict(RESEL_GEMM_SPLIT='9'),
        'fp32 mfma': dict(RESEL_GEMM_SPLIT='0')}
res = {}
for name, e in runs.items():
    r = subprocess.run([sys.executable, '-c', code, rnn, str(N)], capture_output=True, text=True, env=dict(os.environ, **e))
    res[name] = json.loads(r.stdout.strip().splitlines()[-1])


def worst(a, b):
    w, wk = 0.0, None
    for la, lb in zip(a['logs'], b['logs']):
        for k in la:
            d = abs(la[k] - lb[k]) / max(1e-6, abs(lb[k]))
            if d > w:
                w, wk = d, k
    return w, wk


for name, v in res.items():
    last = v['logs'][-1]
    print(f"{name:10s} critic_loss {last['critic_loss']:.6f} actor_loss {last.get('actor_loss', float('nan')):.6f} sum|policy| {v['psum']:.6f} sum|value| {v['vsum']:.6f}")
for name in ('split 6', 'split 9', 'fp32 mfma'):
    w, k = worst(res[name], res['library'])
    print(f'{name:10s} vs library: largest relative difference of any logged scalar over {N} updates {w:.2e} ({k}); '
          f"parameter checksums differ by {abs(res[name]['psum'] - res['library']['psum']) / res['library']['psum']:.2e} / "
          f"{abs(res[name]['vsum'] - res['library']['vsum']) / res['library']['vsum']:.2e}")
w, k = worst(res['split 6'], res['fp32 mfma'])
print(f'split 6 vs fp32 mfma: {w:.2e} ({k})')
