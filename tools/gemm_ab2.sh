#!/bin/bash
# A/B of the two split-GEMM editions on the update's shapes (mode 6 = gemm_bf3.hip, 106 = first edition), then the tests.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "gemm" > gpurun_out/gemm_tests.log 2>&1; tail -4 gpurun_out/gemm_tests.log
python tools/bench_gemm_f32.py 66752 6 > gpurun_out/gemm_v2_m6.txt 2>&1
python tools/bench_gemm_f32.py 66752 106 > gpurun_out/gemm_v1_m6.txt 2>&1
paste -d'\n' gpurun_out/gemm_v2_m6.txt gpurun_out/gemm_v1_m6.txt | grep -v "^$" | cut -c1-200
