#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/recurrent-offpolicy-rl_amd/csrc
for fl in "" "-DEXP_NO_BFLY" "-DEXP_NO_EPI" "-DEXP_NO_PLAIN" "-DEXP_NO_BCSTORE" "-DEXP_NO_BFLY -DEXP_NO_EPI -DEXP_NO_PLAIN -DEXP_NO_BCSTORE"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I. -I../../include $fl -c selective_scan.hip -o build/selective_scan.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A6 "sscan_bwd_kernelILi8ELi4" | grep -E "VGPRs:|AGPRs:" | awk '{printf "%s ", $(NF-1)}'
  hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../offpolicy_rnn/hip/libresel_hip.so
  echo "[$fl] $(cd $R && timeout 200 python tools/prof_sscan.py 2>/dev/null | tail -1)"
done
