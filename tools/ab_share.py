"""Full-size A/B of the shared policy pass: same seeds, 3 updates of the bench configuration, logged scalars side by side."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os, json, numpy as np, torch
sys.path[:0] = [%r, os.path.join(%r, 'recurrent-offpolicy-rl_amd')]
from bench import build_trainer
torch.manual_seed(1); np.random.seed(1)
alg = build_trainer(sys.argv[1], 64, 1024)
torch.manual_seed(2); np.random.seed(2)
out = []
for _ in range(3):
    log = dict(alg.train_one_batch()); alg.grad_num += 1
    out.append({k: float(v[0] if isinstance(v, tuple) else v) for k, v in log.items()})
print(json.dumps(dict(logs=out, psum=float(alg.policy.store.flat.double().abs().sum()))))
''' % (ROOT, ROOT)
rnn = sys.argv[1] if len(sys.argv) > 1 else 'smamba_s32_c16_b2_nln'
res = {}
for flag in ('1', '0'):
    env = dict(os.environ, RESEL_SHARE_POLICY_PASS=flag)
    r = subprocess.run([sys.executable, '-c', code, rnn], capture_output=True, text=True, env=env)
    res[flag] = json.loads(r.stdout.strip().splitlines()[-1])
worst = 0.0
for a, b in zip(res['1']['logs'], res['0']['logs']):
    for k in a:
        d = abs(a[k] - b[k]) / max(1e-6, abs(b[k]))
        worst = max(worst, d)
print('shared', res['1']['logs'][-1]['critic_loss'], res['1']['logs'][-1].get('actor_loss'), res['1']['psum'])
print('two   ', res['0']['logs'][-1]['critic_loss'], res['0']['logs'][-1].get('actor_loss'), res['0']['psum'])
print('largest relative difference of any logged scalar over 3 updates:', worst)
