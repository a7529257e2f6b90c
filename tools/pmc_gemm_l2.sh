#!/bin/bash
# L2 behaviour of one resel_gemm_f32 shape: requests, hits, misses, reads that leave the L2 (rocprofv3 --pmc, one pass per set).
# usage (GPU box): tools/pmc_gemm_l2.sh M N K akc bkc reps split
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_gemm_l2
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SETS=("TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_NC_READ_REQ_sum" "FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE")
for i in 0 1 2 3; do
  timeout 200 rocprofv3 --pmc ${SETS[$i]} -d $OUT -o p$i --output-format csv -- python3 $R/tools/prof_gemm.py "$@" > $OUT/p$i.log 2>&1
  tail -1 $OUT/p$i.log
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in glob.glob('$OUT/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gemm_' in r['Kernel_Name'] and 'fixup' not in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value']); cnt[r['Counter_Name']] += 1
for c in sorted(tot): print('   %-30s %16.0f per launch' % (c, tot[c] / cnt[c]))
PY
