"""Per-aten-op device time by input shape for one config-2 update (GEMM shapes, elementwise tails).  GPU box only."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'recurrent-offpolicy-rl_amd')]
import torch
from collections import defaultdict
from torch.profiler import profile, ProfilerActivity
from bench import build_trainer

rnn = sys.argv[1] if len(sys.argv) > 1 else 'smamba_s32_c16_b2_nln'
alg = build_trainer(rnn, 64, 1024)
for _ in range(2):
    alg.train_one_batch(); alg.grad_num += 1
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    alg.train_one_batch()
    torch.cuda.synchronize()
agg = defaultdict(lambda: [0, 0.0])
for e in prof.key_averages(group_by_input_shape=True):
    agg[(e.key, str(e.input_shapes)[:110])][0] += e.count
    agg[(e.key, str(e.input_shapes)[:110])][1] += e.self_device_time_total
tot = sum(v[1] for v in agg.values())
print(f'total self device time {tot/1e3:.2f} ms')
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f'{v[1]/1e3:8.3f} ms  x{v[0]:3d}  {k[0][:38]:38s} {k[1]}')
