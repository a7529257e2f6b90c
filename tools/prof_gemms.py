"""Per-shape time of the library GEMMs (aten::mm / addmm / bmm / baddbmm) in one update, with achieved TFLOP/s."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'recurrent-offpolicy-rl_amd')]
import torch
from torch.profiler import profile, ProfilerActivity
from collections import defaultdict
from bench import build_trainer
rnn = sys.argv[1] if len(sys.argv) > 1 else 'smamba_s32_c16_b2_nln'
alg = build_trainer(rnn, 64, 1024)
for _ in range(3):
    alg.train_one_batch(); alg.grad_num += 1
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    alg.train_one_batch()
    torch.cuda.synchronize()
agg = defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.name in ('aten::mm', 'aten::addmm', 'aten::bmm', 'aten::baddbmm'):
        agg[(e.name, str(e.input_shapes))][0] += 1
        agg[(e.name, str(e.input_shapes))][1] += e.self_device_time_total
tot = 0.0
for (name, shp), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    sh = eval(shp)
    mats = [s for s in sh if len(s) >= 2]
    a, b = mats[-2], mats[-1]
    flops = 2.0 * a[-2] * a[-1] * b[-1] * (a[0] if len(a) == 3 else 1)
    tot += us
    print(f'{us / 1e3:7.3f} ms x{n:2d} {us / n:8.1f} us  {flops * n / us / 1e6:6.1f} TF/s  {name:13s} {shp}')
print(f'total {tot / 1e3:.2f} ms')
