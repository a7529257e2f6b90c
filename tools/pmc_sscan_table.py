"""Markdown table + derived issue accounting of the two scan kernels from `tools/pmc_sscan.sh` passes (rocprofv3 --pmc csv files).
usage: python tools/pmc_sscan_table.py [gpurun_out/pmc_sscan] > profiles/r06_sscan_pmc.md   (body; the notes are written by hand)"""
import collections, csv, glob, sys
d = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc_sscan'
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(d + '/p*_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        name = 'sscan_fwd2_kernel<8,4,32,0>' if 'sscan_fwd' in k else 'sscan_bwd_kernel<8,4>' if 'sscan_bwd_kernel' in k else None
        if name:
            agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
L = 1043
for name, v in agg.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    print(f'## {name}\n\n| counter | per launch | per wave | share of SQ_WAVE_CYCLES |\n|---|---|---|---|')
    wc, nw = m['SQ_WAVE_CYCLES'], m['SQ_WAVES']
    for c in sorted(m):
        share = f'{100 * m[c] / wc:.1f} %' if c.startswith(('SQ_ACTIVE', 'SQ_WAIT', 'SQ_INST_CYCLES', 'SQ_LDS_BANK', 'SQ_LDS_IDX', 'SQ_INST_LEVEL')) else ''
        print(f'| {c} | {m[c]:.4g} | {m[c] / nw:.5g} | {share} |')
    valu_busy = 2 * m['SQ_ACTIVE_INST_VALU'] / wc
    print(f'\nderived: {m["SQ_INSTS_VALU"] / nw / L:.1f} VALU instructions per wave-step, {4 * m["SQ_ACTIVE_INST_VALU"] / m["SQ_INSTS_VALU"]:.2f} cycles per VALU instruction, '
          f'{4 * wc / nw / L:.0f} cycles per wave-step; VALU busy per SIMD (two waves) {100 * valu_busy:.1f} %, idle {100 * (1 - valu_busy):.1f} %; per wave: '
          f'waiting on s_waitcnt / barrier {100 * m["SQ_WAIT_INST_ANY"] / wc:.1f} % (LDS counter part {100 * m["SQ_WAIT_INST_LDS"] / wc:.1f} %), '
          f'issuing LDS {100 * m["SQ_ACTIVE_INST_LDS"] / wc:.1f} %, scalar {100 * m["SQ_ACTIVE_INST_SCA"] / wc:.1f} %, FLAT / global {100 * m["SQ_ACTIVE_INST_FLAT"] / wc:.1f} %, '
          f'LDS bank-conflict cycles / LDS active cycles {100 * m["SQ_LDS_BANK_CONFLICT"] / max(1, m["SQ_LDS_IDX_ACTIVE"]):.1f} %, '
          f'average LDS instructions in flight per wave {m["SQ_INST_LEVEL_LDS"] / wc:.3f}, VMEM {m["SQ_INST_LEVEL_VMEM"] / wc:.3f}\n')
