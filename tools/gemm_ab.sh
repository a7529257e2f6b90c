#!/bin/bash
# In-situ A/B of the GEMM routing: kernel stats of the bench with the hand-written GEMM on (default) and off.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for mode in on off; do
  if [ $mode = off ]; then export RESEL_GEMM_F32_MIN_ROWS=1000000000; else unset RESEL_GEMM_F32_MIN_ROWS; fi
  OUT=$R/gpurun_out/gemm_ab_$mode; mkdir -p $OUT
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o ab --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-strict-leg > $OUT/log.txt 2>&1
  rm -f $OUT/*_kernel_trace.csv
  python3 - $OUT/ab_kernel_stats.csv $mode <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
cat = {}
for r in rows:
    n = r['Name']; t = float(r['TotalDurationNs']) / 7e6
    k = 'lib gemm' if n.startswith('Cijk') else 'gemm_f32' if 'gemm_f' in n else 'bias_act' if 'bias_act' in n else 'elementwise' if 'elementwise' in n or 'vectorized' in n else 'other'
    cat[k] = cat.get(k, 0) + t
print(sys.argv[2], {k: round(v, 2) for k, v in cat.items()}, 'total', round(sum(cat.values()), 2), 'ms/update')
PY
  tail -1 $OUT/log.txt | cut -c1-140
done
