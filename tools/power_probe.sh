#!/bin/bash
# Board power and clocks (rocm-smi, read-only) sampled while one GEMM shape runs in a loop: is the kernel held at a power cap?
# usage (GPU box): tools/power_probe.sh M N K split
R=${GRAFT_REPO_ROOT:-/root/repo}
python3 $R/tools/prof_gemm.py $1 $2 $3 1 1 6000 $4 > /tmp/probe_gemm.log 2>&1 &
PID=$!
sleep 4
for i in 1 2 3; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (edge|junction|memory)" | head -8
  echo "--"
  sleep 1
done
wait $PID
tail -1 /tmp/probe_gemm.log
