#!/bin/bash
# PMC passes over the stand-alone scan lab (counters only).  usage: pmc_lab.sh <binary> [lab args...]
R=${GRAFT_REPO_ROOT:-/root/repo}
BIN=$1; shift
OUT=$R/gpurun_out/pmc_lab
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SETS=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
      "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
      "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
      "GRBM_GUI_ACTIVE")
for i in 0 1 2 3; do
  timeout 120 rocprofv3 --pmc ${SETS[$i]} -d $OUT -o p$i --output-format csv -- $R/tools/micro/bin/$BIN "$@" > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('$OUT/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][-40:]
        tot[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
for k in tot:
    print('==', k)
    for c in sorted(tot[k]): print('   %-26s %14.0f per launch' % (c, tot[k][c] / cnt[k][c]))
PY
