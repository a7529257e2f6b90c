"""Soak run: N consecutive updates of the bench configuration; prints the log scalars every 20 updates and checks they stay finite.
python tools/soak.py [rnn] [updates] [graph]     third argument 'graph': every update through GraphedUpdate.step() (replays)"""
import sys, os, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'recurrent-offpolicy-rl_amd')]
import torch
from bench import build_trainer
rnn = sys.argv[1] if len(sys.argv) > 1 else 'smamba_s32_c16_b2_nln'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
alg = build_trainer(rnn, 64, 1024)
# SOAK_PER=2: the reference's published cadence (the actor steps on every second update: two graphs alternate); SOAK_CLIP=1: gradient-norm clipping on
alg.parameter.policy_update_per = int(os.environ.get('SOAK_PER', '1'))
if os.environ.get('SOAK_CLIP') == '1':
    alg.parameter.value_max_gradnorm, alg.parameter.policy_max_gradnorm = 10.0, 1.0
step = alg.train_one_batch
if len(sys.argv) > 3 and sys.argv[3] == 'graph':
    from offpolicy_rnn.algorithm.graphed_update import GraphedUpdate
    gu = GraphedUpdate(alg, warmup=1)
    step = gu.step
for i in range(n):
    log = dict(step())
    alg.grad_num += 1
    vals = {k: (v[0] if isinstance(v, tuple) else v) for k, v in log.items()}
    assert all(math.isfinite(float(v)) for v in vals.values()), (i, vals)
    if i % 20 == 0 or i == n - 1:
        print(i, {k: round(float(vals[k]), 4) for k in ('critic_loss', 'actor_loss', 'log_prob', 'log_alpha', 'target_q_max', 'clip_min', 'clip_max') if k in vals})
print('soak ok; device memory reserved %.1f GB, graphs %s' % (torch.cuda.memory_reserved() / 2 ** 30, len(gu.graphs) if len(sys.argv) > 3 and sys.argv[3] == 'graph' else '-'))
