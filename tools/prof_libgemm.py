"""Which library GEMMs (aten::mm / addmm / bmm / baddbmm) does one update still issue, with shapes, device time and call site."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'recurrent-offpolicy-rl_amd')]
import torch
from torch.profiler import profile, ProfilerActivity
from collections import defaultdict
from bench import build_trainer
rnn = sys.argv[1] if len(sys.argv) > 1 else 'smamba_s32_c16_b2_nln'
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 64
algo = sys.argv[3] if len(sys.argv) > 3 else 'sac'
T = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
alg = build_trainer(rnn, rows, T, algo=algo)
for _ in range(2):
    alg.train_one_batch(); alg.grad_num += 1
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    alg.train_one_batch()
    torch.cuda.synchronize()
names = ('aten::mm', 'aten::addmm', 'aten::bmm', 'aten::baddbmm', 'aten::linear', 'aten::matmul', 'aten::_to_copy', 'aten::cat', 'aten::elu', 'aten::elu_backward', 'aten::sum', 'aten::mul', 'aten::add', 'aten::add_', 'aten::copy_', 'aten::fill_', 'aten::zero_')
agg = defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.name in names and e.self_device_time_total > 0:
        st = [s for s in (e.stack or []) if 'offpolicy_rnn' in s][:2]
        key = (e.name, str(e.input_shapes)[:70], ' <- '.join(s.split('/')[-1][:48] for s in st))
        agg[key][0] += 1
        agg[key][1] += e.self_device_time_total
tot = 0
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    tot += v[1]
    print(f'{v[1]/1e3:7.2f} ms x{v[0]:3d} {k[0]:14s} {k[1]:70s} {k[2]}')
print('listed total %.2f ms' % (tot / 1e3))
