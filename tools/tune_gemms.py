"""Regenerate offpolicy_rnn/hip/gemm_tuning/gfx950.csv (see hip/gemm_select.py).

On the GPU box:   python tools/tune_gemms.py run --rnn smamba_s32_c16_b2_nln --rows 64 --horizon 1024 [--algo sac]
                  (three updates with TunableOp searching; winners -> gpurun_out/tuned_<tag>.csv)
                  (--fresh searches every shape again.  Do NOT combine with PYTORCH_TUNABLEOP_ROTATING_BUFFER_SIZE: the one
                  attempt with a 2 GB rotating buffer took the GPU box down.)
Anywhere:         python tools/tune_gemms.py merge      (folds gpurun_out/tuned_*.csv into the tracked table; for a shape
                  seen twice the faster entry wins; validator lines must agree)
"""
import argparse
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLE = os.path.join(ROOT, 'recurrent-offpolicy-rl_amd', 'offpolicy_rnn', 'hip', 'gemm_tuning', 'gfx950.csv')


def run(a):
    tag = f'{a.rnn}_{a.algo}_b{a.rows}_t{a.horizon}' + ('_fresh' if a.fresh else '')
    out = os.path.join(ROOT, 'gpurun_out', f'tuned_{tag}.csv')
    os.makedirs(os.path.dirname(out), exist_ok=True)
    os.environ.update(PYTORCH_TUNABLEOP_ENABLED='1', PYTORCH_TUNABLEOP_TUNING='1', RESEL_GEMM_SELECT='0')
    sys.path[:0] = [ROOT, os.path.join(ROOT, 'recurrent-offpolicy-rl_amd')]
    import torch
    import torch.cuda.tunable as tunable
    from bench import build_trainer
    tunable.enable(True)
    tunable.tuning_enable(True)
    tunable.set_filename(out)
    if os.path.exists(TABLE) and not a.fresh:
        tunable.read_file(TABLE)                     # shapes already in the table are not searched again
    alg = build_trainer(a.rnn, a.rows, a.horizon, algo=a.algo)
    for _ in range(3):
        alg.train_one_batch()
        alg.grad_num += 1
    torch.cuda.synchronize()
    print('wrote', out)


def merge(_):
    validators, best = None, {}
    files = ([TABLE] if os.path.exists(TABLE) else []) + sorted(glob.glob(os.path.join(ROOT, 'gpurun_out', 'tuned_*.csv')))
    for f in files:
        v = [ln.strip() for ln in open(f) if ln.startswith('Validator')]
        if validators is None:
            validators = v
        elif v != validators:
            print('skip (validators differ):', f)
            continue
        for ln in open(f):
            if ln.startswith('Validator') or not ln.strip():
                continue
            op, shape, sol, t = ln.strip().split(',')
            if (op, shape) not in best or float(t) < float(best[(op, shape)][1]):
                best[(op, shape)] = (sol, t)
    with open(TABLE, 'w') as fh:
        fh.write('\n'.join(validators) + '\n')
        for (op, shape), (sol, t) in sorted(best.items()):
            fh.write(f'{op},{shape},{sol},{t}\n')
    print(f'{len(best)} shapes -> {TABLE}')


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest='cmd', required=True)
    r = sub.add_parser('run')
    r.add_argument('--rnn', default='smamba_s32_c16_b2_nln')
    r.add_argument('--algo', default='sac')
    r.add_argument('--rows', type=int, default=64)
    r.add_argument('--horizon', type=int, default=1024)
    r.add_argument('--fresh', action='store_true', help='search every shape again (do not preload the tracked table)')
    sub.add_parser('merge')
    a = ap.parse_args()
    {'run': run, 'merge': merge}[a.cmd](a)
