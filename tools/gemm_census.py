"""Census of the hand-written GEMM calls of one update: shape, operand layouts, call site (file:line of the caller of ops.gemm_f32 and
of its caller), launches and device time (one HIP event pair per call).  Tells which operands are worth a producer-side magnitude."""
import sys, os, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'recurrent-offpolicy-rl_amd')]
import torch
from collections import defaultdict
from bench import build_trainer
from offpolicy_rnn.hip import ops
rnn = sys.argv[1] if len(sys.argv) > 1 else 'smamba_s32_c16_b2_nln'
rows, horizon = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (64, 1024)
alg = build_trainer(rnn, rows, horizon)
for _ in range(3):
    alg.train_one_batch(); alg.grad_num += 1
torch.cuda.synchronize()
real = ops.gemm_f32
rec = []


def hooked(A, B, a_kcontig=True, b_kcontig=True, bias=None, act=None, out=None, split=None, **kw):
    st = traceback.extract_stack(limit=4)
    site = ' <- '.join(f'{os.path.basename(f.filename)}:{f.lineno}' for f in reversed(st[:-1]))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = real(A, B, a_kcontig, b_kcontig, bias, act, out, split, **kw)
    e1.record()
    batch = A.shape[0] if A.dim() == 3 else 1
    M, K = (A.shape[-2], A.shape[-1]) if a_kcontig else (A.shape[-1], A.shape[-2])
    N = B.shape[-2] if b_kcontig else B.shape[-1]
    rec.append(((batch, M, N, K, int(a_kcontig), int(b_kcontig), act, ops.LAST_SPLIT[0], site), e0, e1))
    return r


ops.gemm_f32 = hooked
real_amax = ops.amax
pre = []


def hooked_amax(x):
    st = traceback.extract_stack(limit=5)
    pre.append((tuple(x.shape), ' <- '.join(f'{os.path.basename(f.filename)}:{f.lineno}' for f in reversed(st[:-1]))))
    return real_amax(x)


ops.amax = hooked_amax
import offpolicy_rnn.models.ensemble_linear_model as elm
alg.train_one_batch()
torch.cuda.synchronize()
agg = defaultdict(lambda: [0, 0.0])
for key, e0, e1 in rec:
    agg[key][0] += 1
    agg[key][1] += e0.elapsed_time(e1) * 1e3
tot = sum(v[1] for v in agg.values())
print(f'{len(rec)} calls, {tot / 1e3:.2f} ms (event pairs include launch gaps)')
for key, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    b, M, N, K, akc, bkc, act, mode, site = key
    print(f'{us / 1e3:7.3f} ms x{n:2d} {us / n:7.1f} us  mode {mode} b{b} M{M} N{N} K{K} a{akc} b{bkc} {str(act):5s} {site}')

print(f'{len(pre)} magnitude pre-passes')
cnt = defaultdict(int)
for k in pre:
    cnt[k] += 1
for (shape, site), n in sorted(cnt.items(), key=lambda kv: -kv[1] * (kv[0][0][0] * kv[0][0][-1])):
    print(f'  x{n} {shape} {site}')

# by shape: time against the two floors of the call - three fp16 plane products at the dense fp16 matrix rate (2.5 PFLOP/s) and the
# operand + output bytes at 5 TB/s
print('by shape: ms total | us per call | floor us (matrix, bytes) | calls')
shp = defaultdict(lambda: [0, 0.0])
for (b, M, N, K, akc, bkc, act, mode, site), (n, us) in agg.items():
    shp[(b, M, N, K, akc, bkc, mode)][0] += n
    shp[(b, M, N, K, akc, bkc, mode)][1] += us
fl_tot = 0.0
for (b, M, N, K, akc, bkc, mode), (n, us) in sorted(shp.items(), key=lambda kv: -kv[1][1]):
    t_m = 3 * 2.0 * b * M * N * K / 2.5e9
    t_b = 4.0 * b * (M * K + N * K + M * N) / 5e6
    fl_tot += n * max(t_m, t_b)
    print(f'{us / 1e3:7.3f} | {us / n:7.1f} | {t_m:6.1f} {t_b:6.1f} | x{n:2d} mode {mode} b{b} M{M} N{N} K{K} a{akc} b{bkc}')
print(f'sum of floors {fl_tot / 1e3:.2f} ms of {tot / 1e3:.2f}')
