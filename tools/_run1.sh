RESEL_GEMM_EDITION=3 python3 tools/gemm_census.py 2>/dev/null | sed -n '/by shape/,$p' | head -28 > gpurun_out/c3.txt
RESEL_GEMM_EDITION=2 python3 tools/gemm_census.py 2>/dev/null | sed -n '/by shape/,$p' | head -28 > gpurun_out/c2.txt
paste -d'|' <(cut -c1-17 gpurun_out/c3.txt) <(cut -c1-110 gpurun_out/c2.txt) | head -30
tail -1 gpurun_out/c3.txt; RESEL_GEMM_EDITION=3 python3 tools/gemm_census.py 2>/dev/null | grep "sum of floors"; RESEL_GEMM_EDITION=2 python3 tools/gemm_census.py 2>/dev/null | grep "sum of floors"
