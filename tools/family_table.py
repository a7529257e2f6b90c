"""Kernel time per update by kernel family for profile tags under profiles/ (rocprofv3 --stats summaries of `bench.py`, tools/profile_bench.sh):
usage: family_table.py r04f r05f r05h"""
import csv, re, collections, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAM = [('selective scan (fwd, bwd, reduce, parameter gradients)', r'sscan'),
       ('GEMM mode 2, producer / consumer edition', r'gemm_ws_kernel<(true|false), (true|false), 2(, 0)?>'),
       ('GEMM fused epilogues (`head`, `dact`)', r'gemm_ws_kernel<[^>]*, [45]>'),
       ('GEMM other forms (mode 6, fp32 MFMA, second edition)', r'gemm_bf3_kernel|gemm_f32_kernel|gemm_w8'),
       ('GEMM fix-ups', r'fixup'), ('conv1d', r'conv_'), ('LayerNorm', r'ln_(fwd|bwd)'),
       ('bias / activation / head passes', r'bias_act|head_(fwd|bwd)|head_fold|dact_tail'), ('column sums', r'colsum'),
       ('magnitude pre-passes', r'amax'), ('ATen cat / copies / fills / element-wise', r'at::native|rocclr'), ('everything else', r'.')]
tags = sys.argv[1:]
res = {}
for tag in tags:
    rows = list(csv.DictReader(open(os.path.join(ROOT, 'profiles', f'{tag}_kernel_stats.csv'))))
    b = json.loads(open(os.path.join(ROOT, 'profiles', f'{tag}_bench.json')).read())
    n = b['steps'] + b['warmup']
    acc = collections.OrderedDict((f[0], [0.0, 0.0]) for f in FAM)
    for r in rows:
        for name, pat in FAM:
            if re.search(pat, r['Name']):
                acc[name][0] += float(r['TotalDurationNs']) / n / 1e6
                acc[name][1] += int(r['Calls']) / n
                break
    res[tag] = (acc, b)
print('| kernel family | ' + ' | '.join(f'{t}: ms / update (launches)' for t in tags) + ' |')
print('|---|' + '---|' * len(tags))
for name, _ in FAM:
    print(f'| {name} | ' + ' | '.join(f'{res[t][0][name][0]:.2f} ({res[t][0][name][1]:.0f})' for t in tags) + ' |')
print('| **all kernels** | ' + ' | '.join(f'**{sum(v[0] for v in res[t][0].values()):.2f}** ({sum(v[1] for v in res[t][0].values()):.0f})' for t in tags) + ' |')
print('| eager ms per update of that run | ' + ' | '.join(f"{res[t][1]['ms_per_step']:.2f}" for t in tags) + ' |')
