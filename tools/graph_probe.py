"""Eager update vs the hipGraph replay of the whole update (algorithm/graphed_update.py).  python tools/graph_probe.py [rnn] [rows] [horizon]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'recurrent-offpolicy-rl_amd')]
import torch
from bench import build_trainer
from offpolicy_rnn.algorithm.graphed_update import GraphedUpdate
rnn = sys.argv[1] if len(sys.argv) > 1 else 'smamba_s32_c16_b2_nln'
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 8
T = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
alg = build_trainer(rnn, rows, T)
alg.defer_log = True


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn(); alg.grad_num += 1
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


timed(alg.train_one_batch, 5)
print('eager ms/update', timed(alg.train_one_batch, 20))
g = GraphedUpdate(alg)
timed(g.step, 3)
print('graph replay ms/update', timed(g.step, 20), 'graphs', len(g.graphs), 'eager fallbacks', g.eager_fallbacks)
