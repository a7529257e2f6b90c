"""BASELINE.md section 3 step 1 / SURVEY.md 8(d): wall time of the REFERENCE's own GRU SAC-REDQ trainer against the repo's CPU
restatement (oracle/trainer.py, the `cpu_baseline` leg of bench.py) on the same synthetic workload, in THIS container (the
Python reference cannot travel to the GPU box).  One warm-up update, then `updates` timed ones, for each.  Build container
only:  python tools/cpu_ratio_reference_vs_oracle.py [B=8] [T=128] [threads=8] [updates=3]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, T, threads, updates = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 8), (2, 128), (3, 8), (4, 3)))
import numpy as np
import torch
torch.set_num_threads(threads)

# ---- the repo's restatement (imported first: the reference harness below replaces the `offpolicy_rnn` module name)
sys.path.insert(0, ROOT)
from oracle.trainer import time_cpu_baseline
r_or = time_cpu_baseline('gru', B=B, T=T, updates=updates, warmup=1, threads=threads)

# ---- the reference itself, through the stub harness of tests/golden/generate_golden.py
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import generate_golden as G
OBS, ACT = 17, 6
G.ENV = dict(obs=OBS, act=ACT, T=T)
G.install_stubs(OBS, ACT, T)
from offpolicy_rnn.algorithm.sac_full_length_rnn_redq_sep_optim import SACFullLengthRNNREDQ_SEP_OPTIM
from offpolicy_rnn.buffers.transition_buffer.replay_memory import Transition
torch.manual_seed(0)
np.random.seed(0)
par = G.make_parameter('gru', D=256, sac_batch_size=B * T - 1, max_buffer_transition_num=4 * B * T, policy_embedding_dim=128,
                       value_embedding_dim=128, policy_uni_model_input_mapping_dim=128, value_uni_model_input_mapping_dim=128)
alg = SACFullLengthRNNREDQ_SEP_OPTIM(par)
rs = np.random.RandomState(0)
for _ in range(2 * B):
    o, a, r = G.synth_traj(rs, T, OBS, ACT)
    G.push_traj(alg.replay_buffer, Transition, o, a, r, early_done=False)
alg.train_one_batch()
t0 = time.time()
n = 0
for _ in range(updates):
    n += alg.train_one_batch()['real_batch_size']
    alg.grad_num += 1
dt_ref = (time.time() - t0) / updates
print(f'reference GRU SAC-REDQ trainer : {dt_ref:.3f} s/update  ({n / updates / dt_ref:.0f} env-steps/s) at B={B}, T={T}, {threads} threads')
print(f'oracle restatement (kind: port): {r_or["seconds_per_update"]:.3f} s/update  ({r_or["value"]:.0f} env-steps/s)')
print(f'ratio restatement / reference  : {r_or["seconds_per_update"] / dt_ref:.2f}')
