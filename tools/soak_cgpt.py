"""Soak run of BASELINE configs[2] (cgpt TD3, 32 rows, dropout on): N consecutive updates, every logged scalar finite.
python tools/soak_cgpt.py [n] [graph]     second argument 'graph': every update through GraphedUpdate.step() (replays, device-side dropout base)"""
import sys, os, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'recurrent-offpolicy-rl_amd')]
import torch
from bench import build_trainer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
alg = build_trainer('cgpt_h8_l6_p0.1_ml1024_rms', 32, 1024, algo='td3')
step = alg.train_one_batch
if len(sys.argv) > 2 and sys.argv[2] == 'graph':
    from offpolicy_rnn.algorithm.graphed_update import GraphedUpdate
    gu = GraphedUpdate(alg, warmup=1)
    step = gu.step
for i in range(n):
    log = dict(step())
    alg.grad_num += 1
    vals = {k: (v[0] if isinstance(v, tuple) else v) for k, v in log.items()}
    assert all(math.isfinite(float(v)) for v in vals.values()), (i, vals)
    if i % 25 == 0 or i == n - 1:
        print(i, {k: round(float(vals[k]), 4) for k in ('critic_loss', 'actor_loss', 'target_q_max', 'clip_min', 'clip_max') if k in vals})
print('soak ok')
