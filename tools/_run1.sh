python bench.py --no-suite --no-cpu-baseline --no-rccl-leg --no-strict-leg 2>gpurun_out/err.txt > gpurun_out/b.json; python -c "
import json; j=json.loads(open('gpurun_out/b.json').read().strip().splitlines()[-1]); print('default', round(j['ms_per_step'],3), j['launch'], j['eager_ms_per_step'], j['sscan'], j['roofline']['avg_us'], j['roofline']['frac'])" || tail -5 gpurun_out/err.txt
python bench.py --no-suite --no-cpu-baseline --no-rccl-leg --no-strict-leg --no-graph-update 2>gpurun_out/err.txt | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('eager', round(j['ms_per_step'],3), j['launch'], j['graph_update_leg'])" || tail -5 gpurun_out/err.txt
python -m pytest tests/test_data_parallel_gpu.py tests/test_trainer_gpu.py -m gpu -q -x 2>&1 | tail -4
