"""Time the oracle trainer (bench.py's cpu_baseline leg) on this host: python tools/cpu_baseline_probe.py <rnn> <B> <T> <threads>."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from oracle.trainer import time_cpu_baseline
rnn, B, T, th = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
t = time.time()
r = time_cpu_baseline(rnn, B=B, T=T, updates=1, warmup=0, threads=th)
print(r, 'wall', round(time.time() - t, 1))
