echo "== eager, default products"; python3 tools/soak.py smamba_s32_c16_b2_nln 300 2>&1 | grep -v amdgpu.ids | tail -4
echo "== graph, default products"; python3 tools/soak.py smamba_s32_c16_b2_nln 300 graph 2>&1 | grep -v amdgpu.ids | tail -4
echo "== eager, mode 6"; RESEL_GEMM_SPLIT=6 python3 tools/soak.py smamba_s32_c16_b2_nln 300 2>&1 | grep -v amdgpu.ids | tail -3
