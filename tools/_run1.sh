timeout 2000 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_default.json 2>gpurun_out/err.txt; tail -1 gpurun_out/err.txt
