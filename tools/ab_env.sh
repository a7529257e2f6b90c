#!/bin/bash
# Same-box A/B of the bench line under environment settings: ab_env.sh "VAR=a" "VAR=b" ... (each run twice, interleaved)
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for setting in "$@"; do
    out=$(env $setting python $R/bench.py --no-suite --no-cpu-baseline --no-rccl-leg --no-strict-leg 2>/dev/null | tail -n 1)
    echo "$setting | $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("replayed %.3f ms  eager %.3f ms" % (d["ms_per_step"], d["eager_ms_per_step"]))')"
  done
done
