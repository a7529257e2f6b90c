"""Probe: can one whole train_one_batch() be captured in a hipGraph (torch.cuda.graph) and what does a replay cost?
python tools/graph_probe.py [rnn] [rows] [horizon]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'recurrent-offpolicy-rl_amd')]
import torch
from bench import build_trainer
rnn = sys.argv[1] if len(sys.argv) > 1 else 'smamba_s32_c16_b2_nln'
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 8
T = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
alg = build_trainer(rnn, rows, T)
alg.defer_log = True
for _ in range(5):
    alg.train_one_batch(); alg.grad_num += 1
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    alg.train_one_batch(); alg.grad_num += 1
torch.cuda.synchronize()
print('eager ms/update', (time.perf_counter() - t0) / 10 * 1e3)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
try:
    with torch.cuda.stream(s):
        for _ in range(2):
            alg.train_one_batch(); alg.grad_num += 1
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        log = alg.train_one_batch()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    print('graph replay ms/update', (time.perf_counter() - t0) / 20 * 1e3)
except Exception as e:
    import traceback
    traceback.print_exc()
    print('capture failed:', repr(e)[:300])
