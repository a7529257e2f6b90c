export RESEL_HIP_LIBRARY=tools/micro/bin/libresel_ws_asm.so
RESEL_GEMM_EDITION=3 timeout 300 python3 tools/_chk.py 2>&1 | grep -v amdgpu.ids | awk '{ if ($4+0 > 1e-5 || $4=="nan") print "BAD", $0; else n++ } END { print n, "shapes ok" }'
RESEL_GEMM_EDITION=3 timeout 600 python3 -m pytest tests/test_hip_ops.py -m gpu -x -q -k "gemm" 2>&1 | tail -2
echo "== edition 3"; RESEL_GEMM_EDITION=3 timeout 300 python3 tools/ab_f16x3.py 2>&1 | grep -v amdgpu.ids | head -6 | cut -c1-110
echo "== edition 2"; RESEL_GEMM_EDITION=2 python3 tools/ab_f16x3.py 2>&1 | grep -v amdgpu.ids | head -6 | cut -c1-110
