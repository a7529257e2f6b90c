cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 20 --warmup 5 --no-suite --no-rccl-leg --no-cpu-baseline --no-strict-leg $2 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', round(d['ms_per_step'],2))
"; }
RESEL_GEMM_F32_MIN_K=4 run k4
RESEL_GEMM_F32_MIN_DIM=12 RESEL_GEMM_F32_MIN_K=4 run dim12_k4
RESEL_GEMM_F32_MIN_DIM=16 RESEL_GEMM_F32_MIN_K=4 run dim16_k4
RESEL_GEMM_F32_MIN_DIM=32 RESEL_GEMM_F32_MIN_K=4 run dim32_k4
RESEL_GEMM_F32_MIN_K=16 run k16
RESEL_GEMM_F32_MIN_K=4 run k4
run base_b8 "--rows 8"
RESEL_GEMM_F32_MIN_K=4 RESEL_GEMM_F32_MIN_ROWS=1024 run k4_rows1024_b8 "--rows 8"
RESEL_GEMM_F32_MIN_K=4 run k4_b8 "--rows 8"
