"""Where does a GraphedUpdate.step() spend its wall time: the host half (`_prepare`), the replay on the GPU (event pair), the log read-back."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'recurrent-offpolicy-rl_amd')]
import torch
from bench import build_trainer
from offpolicy_rnn.algorithm.graphed_update import GraphedUpdate
rnn = sys.argv[1] if len(sys.argv) > 1 else 'cgpt_h8_l6_p0.1_ml1024_rms'
algo = sys.argv[2] if len(sys.argv) > 2 else 'td3'
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 32
alg = build_trainer(rnn, rows, 1024, algo=algo)
gu = GraphedUpdate(alg, warmup=1)
for _ in range(5):
    gu.step(); alg.grad_num += 1
torch.cuda.synchronize()
real_prepare = gu._prepare
tp = []
def timed_prepare():
    t = time.perf_counter(); k = real_prepare(); tp.append(time.perf_counter() - t); return k
gu._prepare = timed_prepare
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tg, tw = [], []
for _ in range(8):
    t = time.perf_counter()
    e0.record(); gu.step(); e1.record(); alg.grad_num += 1
    torch.cuda.synchronize()
    tw.append(time.perf_counter() - t); tg.append(e0.elapsed_time(e1))
print(f'{rnn}: step wall {1e3 * sum(tw) / len(tw):.2f} ms (synchronised per step), of which _prepare {1e3 * sum(tp) / len(tp):.2f} ms on the host; '
      f'event pair around step() {sum(tg) / len(tg):.2f} ms; graphs {len(gu.graphs)} eager {gu.eager_fallbacks}')
gu.close()
e0.record()
for _ in range(5):
    alg.train_one_batch(); alg.grad_num += 1
e1.record(); torch.cuda.synchronize()
print(f'eager: {e0.elapsed_time(e1) / 5:.2f} ms per update (event pair over 5)')
