"""Host-side cost of an update: time until the 10th train_one_batch() call returns (launch queue still draining) against
the synchronised time - tells whether the update is GPU-bound or launch-bound."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'recurrent-offpolicy-rl_amd')]
import torch
from bench import build_trainer
rnn = sys.argv[1] if len(sys.argv) > 1 else 'smamba_s32_c16_b2_nln'
alg = build_trainer(rnn, 64, 1024)
alg.defer_log = True
for _ in range(3):
    alg.train_one_batch(); alg.grad_num += 1
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    alg.train_one_batch(); alg.grad_num += 1
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'{rnn}: host returns after {1e2 * (t1 - t0):.1f} ms/update, synchronised {1e2 * (t2 - t0):.1f} ms/update')
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    alg.train_one_batch(); alg.grad_num += 1
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(28)
