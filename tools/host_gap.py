"""Eager update vs the same update replayed from one hipGraph vs the sum of the kernel times: what the launch path costs on this box."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'recurrent-offpolicy-rl_amd')]
import torch
from bench import build_trainer
rnn = sys.argv[1] if len(sys.argv) > 1 else 'smamba_s32_c16_b2_nln'
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 64
algo = sys.argv[3] if len(sys.argv) > 3 else 'sac'
alg = build_trainer(rnn, rows, 1024, algo=algo)
alg.defer_log = True


def timed(fn, n=10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
        alg.grad_num += 1
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


for _ in range(3):
    alg.train_one_batch(); alg.grad_num += 1
print('eager            ', round(timed(alg.train_one_batch), 3), 'ms / update')
# host time alone: launch everything without waiting for the GPU in between (the queue absorbs it) vs a CPU-side clock
torch.cuda.synchronize()
t0 = time.perf_counter()
alg.train_one_batch()
host = 1e3 * (time.perf_counter() - t0)
torch.cuda.synchronize()
print('host side of one eager update (returns before the GPU is done)', round(host, 3), 'ms')
from offpolicy_rnn.algorithm.graphed_update import GraphedUpdate
if GraphedUpdate.refusal(alg):
    print('no graph:', GraphedUpdate.refusal(alg)); sys.exit(0)
gu = GraphedUpdate(alg, warmup=1)
for _ in range(4):
    gu.step(); alg.grad_num += 1
print('hipGraph replay  ', round(timed(gu.step), 3), 'ms / update  graphs', len(gu.graphs))
