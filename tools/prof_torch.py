import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'recurrent-offpolicy-rl_amd')]
import torch
from torch.profiler import profile, ProfilerActivity
from bench import build_trainer
alg = build_trainer('smamba_s32_c16_b2_nln', 64, 1024)
for _ in range(2):
    alg.train_one_batch(); alg.grad_num += 1
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    alg.train_one_batch()
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.name in ('aten::copy_', 'aten::contiguous', 'aten::clone', 'aten::fill_', 'aten::zero_', 'aten::add_', 'aten::add', 'aten::sum', 'aten::cat', 'aten::mul', 'aten::elu', 'aten::elu_backward', 'aten::sub', 'aten::neg', 'aten::div')]
from collections import Counter, defaultdict
agg = defaultdict(lambda: [0, 0.0])
for e in evs:
    st = [s for s in (e.stack or []) if 'offpolicy_rnn' in s or 'torch/autograd' in s][:3]
    key = (e.name, str(e.input_shapes)[:80], ' <- '.join(s.split('/')[-1][:60] for s in st))
    agg[key][0] += 1
    agg[key][1] += e.self_device_time_total
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f'{v[1]/1e3:8.2f} ms  x{v[0]:3d}  {k[0]:18s} {k[1]:80s} {k[2]}')
