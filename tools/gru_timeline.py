"""Timeline of one gru update from a rocprofv3 --kernel-trace csv: when do the persistent recurrences run, what overlaps them, where do the
GEMMs of the other streams wait?   usage: python tools/gru_timeline.py <kernel_trace.csv> [update index]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
ev = [(r['Kernel_Name'], (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3, r.get('Queue_Id', r.get('Stream_Id', '?'))) for r in rows]
# updates are delimited by the gather kernel of the device replay
starts = [i for i, e in enumerate(ev) if 'gather_segments_kernel' in e[0]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) - 2
a, b = starts[k], starts[k + 1]
u = ev[a:b]
base = u[0][1]
print(f'update {k}: {len(u)} kernels, {u[-1][2] - base:.0f} us from first start to last end')
short = lambda n: ('GRU_FWD' if 'gru_fwd_persistent' in n else 'GRU_BWD' if 'gru_bwd_persistent' in n else 'gemm' if 'gemm' in n else n.split('(')[0].split('::')[-1][:28])
busy = collections.Counter()
for n, s, e, q in u:
    busy[short(n)] += e - s
print('kernel time by family (us):', {k: round(v) for k, v in busy.most_common(8)})
print('\nrecurrences and the products that start while one is running (start, end, duration in us, queue):')
rec = [(s, e) for n, s, e, q in u if 'persistent' in n]
for n, s, e, q in u:
    sh = short(n)
    inside = any(rs < s < re for rs, re in rec)
    if sh.startswith('GRU') or (sh == 'gemm' and inside and e - s > 150):
        print(f'  {s - base:9.0f} {e - base:9.0f} {e - s:8.0f}  q{q}  {sh}')
