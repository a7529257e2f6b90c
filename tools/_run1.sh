python -m pytest tests -m gpu -q -x 2>&1 | tail -4
for i in 1 2; do python bench.py --no-suite --no-cpu-baseline --no-rccl-leg --no-strict-leg 2>gpurun_out/err.txt | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['ms_per_step'],3), round(j['eager_ms_per_step'],3), j['sscan']['fwd_us'], j['sscan']['bwd_us'], j['kernels']['gemm_f32_kernel'])" || tail -5 gpurun_out/err.txt; done
python tools/gemm_census.py 2>&1 | grep "mode 6\|mode 0\|calls" | head -12
