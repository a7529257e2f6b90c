import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from oracle.trainer import time_cpu_baseline
thr, B, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
t0 = time.time()
r = time_cpu_baseline('gru', B=B, T=T, updates=1, warmup=0, threads=thr)
r['wall_incl_setup'] = time.time() - t0
print(json.dumps(r), flush=True)
