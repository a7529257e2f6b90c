RESEL_GEMM_EDITION=3 timeout 600 python3 -m pytest tests/test_hip_ops.py -m gpu -x -q -k "gemm" 2>&1 | tail -3
echo "== edition 3 touch 4"; RESEL_GEMM_EDITION=3 timeout 300 python3 tools/ab_f16x3.py 2>&1 | grep -v amdgpu.ids
for t in 0 2 8; do echo "== edition 3 touch $t"; RESEL_HIP_LIBRARY=tools/micro/bin/libresel_bf3_T$t.so RESEL_GEMM_EDITION=3 timeout 300 python3 tools/ab_f16x3.py 2>&1 | grep -v amdgpu.ids; done
echo "== edition 2"; RESEL_GEMM_EDITION=2 python3 tools/ab_f16x3.py 2>&1 | grep -v amdgpu.ids
