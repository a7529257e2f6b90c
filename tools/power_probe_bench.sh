#!/bin/bash
# Board power and shader clock (rocm-smi, read-only) sampled every 0.5 s while `bench.py` runs its updates.
# usage (GPU box): tools/power_probe_bench.sh [bench args...]
R=${GRAFT_REPO_ROOT:-/root/repo}
python3 $R/bench.py --steps 900 --warmup 3 --no-cpu-baseline --no-strict-leg "$@" > /tmp/probe_bench.log 2>&1 &
PID=$!
sleep 16
for i in $(seq 1 8); do
  /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ' '
  echo
  sleep 0.5
done
wait $PID
tail -1 /tmp/probe_bench.log | cut -c1-160
