#!/bin/bash
# SQ counters of the attention kernels (tools/prof_attn.py), one rocprofv3 --pmc pass per set.  GPU box.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_attn
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SETS=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
      "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VALU_TRANS SQ_INSTS_MFMA SQ_WAIT_INST_LDS" \
      "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES")
for i in 0 1 2; do
  timeout 200 rocprofv3 --pmc ${SETS[$i]} -d $OUT -o p$i --output-format csv -- python3 $R/tools/${ATTN_PROG:-prof_attn.py} ${ATTN_ARGS} > $OUT/p$i.log 2>&1
done
tail -1 $OUT/p0.log
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('$OUT/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'attn_' not in k: continue
        k = k.split('attn_')[1][:28]
        tot[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
for k in tot:
    print('==', k)
    for c in sorted(tot[k]): print('   %-26s %14.0f per launch' % (c, tot[k][c] / cnt[k][c]))
PY
