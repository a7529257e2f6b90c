for ed in 3 2; do echo "== edition $ed"; RESEL_GEMM_EDITION=$ed bash tools/pmc_gemm.sh 66752 2048 384 1 1 5 2 2>&1 | grep -v "amdgpu.ids\|rocprofv3\|simple_timer"; done
