"""cProfile of the host side of N updates at a small batch (host-bound regime): top functions by own time."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'recurrent-offpolicy-rl_amd')]
import numpy as np, torch
from bench import build_trainer
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(1); np.random.seed(1)
alg = build_trainer('smamba_s32_c16_b2_nln', rows, 1024)
for _ in range(5):
    alg.train_one_batch(); alg.grad_num += 1
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    alg.train_one_batch(); alg.grad_num += 1
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(22)
st.sort_stats('cumtime').print_stats('offpolicy_rnn', 30)
