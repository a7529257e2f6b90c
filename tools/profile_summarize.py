"""Turn gpurun_out/prof_<tag>/ (tools/profile_bench.sh) into the tracked artefacts under profiles/:
   <tag>_kernel_stats.csv (rocprofv3 --stats summary, verbatim), <tag>_bench.json, <tag>_summary.md and traffic.json
   (HBM bytes per launch of the hand-written scan kernels: FETCH_SIZE [KB] x 1024 x 2 - the gfx950 correction for wide
   coalesced reads, MI355X_MICROARCH.md 'HBM' - and WRITE_SIZE [KB] x 1024; separate --pmc passes)."""
import collections, csv, json, os, shutil, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, 'gpurun_out', f'prof_{tag}'), os.path.join(ROOT, 'profiles')
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, f'{tag}_kernel_stats.csv'), os.path.join(dst, f'{tag}_kernel_stats.csv'))
TRAFFIC_NAME = sys.argv[2] if len(sys.argv) > 2 else 'traffic.json'      # secondary workloads keep their own traffic file
shutil.copy(os.path.join(src, f'{tag}_bench.json'), os.path.join(dst, f'{tag}_bench.json'))
bench = json.loads(open(os.path.join(src, f'{tag}_bench.json')).read())
n_upd = bench['steps'] + bench['warmup']
assert not bench.get('graph_update') and not bench.get('suite') and not bench.get('rccl_one_rank_leg'), \
    'profile the eager update alone: --no-graph-update --no-suite --no-rccl-leg --no-strict-leg (the traces are divided by steps + warmup)'


def pmc(name):
    agg = collections.defaultdict(list)
    f = os.path.join(src, f'{tag}_{name}_counter_collection.csv')
    if not os.path.exists(f):
        return {}
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        for short, pat in (('sscan_fwd_kernel', 'sscan_fwd'), ('sscan_bwd_kernel', 'sscan_bwd_kernel'), ('conv_fwd_kernel', 'conv_fwd_kernel'),
                           ('conv_bwd_kernel', 'conv_bwd'), ('ln_fwd_kernel', 'ln_fwd_kernel'), ('ln_bwd_kernel', 'ln_bwd_kernel'),
                           ('attn_fwd_kernel', 'attn_q_kernel<32, 0'), ('attn_dq_kernel', 'attn_q_kernel<32, 1'), ('attn_dkv_kernel', 'attn_dkv_kernel'),
                           ('linrec_real_fwd_kernel', 'linrec_real_fwd'), ('linrec_real_bwd_kernel', 'linrec_real_bwd'),
                           ('linrec_complex_fwd_kernel', 'linrec_complex_fwd'), ('linrec_complex_bwd_kernel', 'linrec_complex_bwd'),
                           ('gemm_f32_kernel', 'gemm_f32_kernel'), ('gemm_f32_kernel', 'gemm_bf3_kernel'), ('gemm_f32_kernel', 'gemm_ws_kernel'), ('gru_fwd_kernel', 'gru_fwd'),
                           ('gru_bwd_kernel', 'gru_bwd')):
            if pat in k:
                agg[short].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def pmc_clusters(name, pat):
    """Distinct per-launch values of one kernel's counter (launches of the forward scan with and without checkpoints differ 5x)."""
    f = os.path.join(src, f'{tag}_{name}_counter_collection.csv')
    if not os.path.exists(f):
        return {}
    c = collections.Counter(round(float(r['Counter_Value']) * 1024 / 1e6, 1) for r in csv.DictReader(open(f)) if pat in r['Kernel_Name'])
    return {f'{k} MB': n for k, n in sorted(c.items())}


fetch, write = pmc('fetch'), pmc('write')
traffic = {}
# FETCH_SIZE counts wide coalesced reads (8 / 16 bytes per lane) at half their bytes on gfx950 (MI355X_MICROARCH.md 'HBM'): x 2.  The
# linear-recurrence scans read ONE dword per lane: calibrated on their own byte count (two passes over v and f + one store of h =
# 5 element passes = 164.1 MB at B 16, T' 2003, C 256; raw FETCH + WRITE = 164.7 MB), the raw counter is exact for them: x 1.
for k in sorted(set(fetch) | set(write)):
    rd, wr = fetch.get(k, 0.0) * 1024 * (1 if k.startswith('linrec_') else 2), write.get(k, 0.0) * 1024
    traffic[k] = {'read_bytes': rd, 'write_bytes': wr, 'total': rd + wr}
import bench as _bench
cfgw = bench['config']['workload']
json.dump({'source': f'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over `python3 bench.py` ({tag}); FETCH_SIZE KB x 1024 x 2 (gfx950), WRITE_SIZE KB x 1024',
           'kernel_source_stamp': _bench.kernel_source_stamp(), 'rnn': cfgw.split(' ')[0], 'rows': bench['config']['global_rows'] // bench['n_gpus'],
           'per_launch_bytes': {k: v['total'] for k, v in traffic.items()}, 'detail': traffic,
           # forward scan: launches without a graph write the output only, launches of a pass that will be differentiated also write
           # the state checkpoints (one every 8 steps) - the distinct per-launch WRITE_SIZE values and how many launches had each
           'sscan_fwd_write_mb_by_launch_kind': pmc_clusters('write', 'sscan_fwd')},
          open(os.path.join(dst, TRAFFIC_NAME), 'w'), indent=1)
rows = list(csv.DictReader(open(os.path.join(src, f'{tag}_kernel_stats.csv'))))
tot = sum(float(r['TotalDurationNs']) for r in rows)
gemm = sum(float(r['TotalDurationNs']) for r in rows if r['Name'].startswith('Cijk'))
own = sum(float(r['TotalDurationNs']) for r in rows if any(k in r['Name'] for k in ('gemm_f32_kernel', 'gemm_bf3_kernel', 'gemm_ws_kernel', 'gemm_bf16_kernel', 'gemm_fixup', 'gemm_bf3_fixup', 'gemm_bf16_fixup')))
with open(os.path.join(dst, f'{tag}_summary.md'), 'w') as fh:
    fh.write(f'# {tag}: rocprofv3 --kernel-trace --stats over `python3 bench.py` ({n_upd} updates incl. warm-up)\n\n')
    fh.write(f'bench line: {bench["value"]:.0f} {bench["unit"]}, {bench["ms_per_step"]:.2f} ms/update; GPU kernel time {tot / n_upd / 1e6:.2f} ms/update, '
             f'of which the hand-written GEMMs (resel_gemm_f32 / resel_gemm_bf16 + their fix-ups) {own / n_upd / 1e6:.2f} ms and library GEMMs (rocBLAS/hipBLASLt fp32) {gemm / n_upd / 1e6:.2f} ms\n\n| kernel | calls | ms/update | avg us |\n|---|---|---|---|\n')
    for r in rows[:30]:
        fh.write(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['TotalDurationNs']) / n_upd / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.1f} |\n")
    fh.write('\n## HBM traffic per launch (PMC)\n\n| kernel | read MB | write MB |\n|---|---|---|\n')
    for k, v in traffic.items():
        fh.write(f"| {k} | {v['read_bytes'] / 1e6:.1f} | {v['write_bytes'] / 1e6:.1f} |\n")
print(open(os.path.join(dst, f'{tag}_summary.md')).read())
