"""Aggregate rocprofv3 --pmc csv files (gpurun_out/pmc_sscan/p*_counter_collection.csv) per kernel."""
import collections, csv, glob, sys
d = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc_sscan'
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(d + '/p*_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        name = ('sscan_fwd' if 'sscan_fwd' in k else 'sscan_bwd' if 'sscan_bwd' in k else k.split('(')[0].split('::')[-1][:40] + ('<fwd>' if ', 0>' in k else '<dq>' if ', 1>' in k else ''))
        agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    print(k)
    for c, vals in v.items():
        print(f'   {c:24s} n={len(vals):3d} mean={sum(vals) / len(vals):.4g}')
