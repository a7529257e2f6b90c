"""Which ATen ops does one update still issue, from where?  A TorchDispatchMode counts every op (autograd's backward ops included) by
name, shapes and the innermost offpolicy_rnn frame that led to it."""
import sys, os, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'recurrent-offpolicy-rl_amd')]
import torch
from collections import Counter
from torch.utils._python_dispatch import TorchDispatchMode
from bench import build_trainer
rnn = sys.argv[1] if len(sys.argv) > 1 else 'smamba_s32_c16_b2_nln'
rows, horizon = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (64, 1024)
alg = build_trainer(rnn, rows, horizon)
for _ in range(3):
    alg.train_one_batch(); alg.grad_num += 1
torch.cuda.synchronize()
cnt = Counter()
SKIP = ('aten.view', 'aten._unsafe_view', 'aten.reshape', 'aten.slice', 'aten.select', 'aten.expand', 'aten.detach', 'aten.t.', 'aten.transpose',
        'aten.unsqueeze', 'aten.squeeze', 'aten.as_strided', 'aten.alias', 'aten.permute', 'aten.unbind', 'aten.split', 'aten.empty', 'aten._local_scalar',
        'aten.is_pinned', 'aten.lift_fresh', 'aten.narrow', 'aten.chunk', 'aten.unflatten', 'aten.flatten', 'aten.contiguous', 'aten._to_copy.default_noop')


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            shapes = tuple(tuple(a.shape) for a in args if isinstance(a, torch.Tensor))[:2]
            site = ''
            for fr in reversed(traceback.extract_stack(limit=40)):
                if 'offpolicy_rnn' in fr.filename and 'small_ops' not in fr.filename:
                    site = f'{os.path.basename(fr.filename)}:{fr.lineno}'
                    break
            cnt[(name, shapes, site)] += 1
        return func(*args, **(kwargs or {}))


with Log():
    alg.train_one_batch()
torch.cuda.synchronize()
tot = sum(cnt.values())
print(f'{tot} ATen ops that launch work in one update (views skipped); by count:')
by_site = Counter()
for (name, shapes, site), n in cnt.items():
    by_site[site] += n
for site, n in by_site.most_common(25):
    print(f'{n:5d}  {site or "(autograd engine / no package frame)"}')
print('--- detail, top 45')
for (name, shapes, site), n in cnt.most_common(45):
    print(f'{n:4d}  {name:34s} {str(shapes)[:60]:60s} {site}')
