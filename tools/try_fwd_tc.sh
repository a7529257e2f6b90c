#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/recurrent-offpolicy-rl_amd/csrc
for tc in 32 28 24 20 16; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I. -I../../include -DSSCAN_FWD_TC=$tc -c selective_scan.hip -o build/selective_scan.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../offpolicy_rnn/hip/libresel_hip.so
  echo "TC=$tc $(cd $R && timeout 200 python tools/prof_sscan.py 2>/dev/null | tail -1)"
done
