python -m pytest tests/test_trainer_gpu.py -m gpu -q -x -k "graphed" -s 2>&1 | grep -v "^$" | tail -12
python bench.py --rnn gru --steps 5 --warmup 3 --no-cpu-baseline --no-strict-leg --no-rccl-leg --no-suite > gpurun_out/bgru.out 2>gpurun_out/bgru.err; echo rc=$?; python -c "
import json; j=json.loads(open('gpurun_out/bgru.out').read().strip().splitlines()[-1]); print('gru', j['ms_per_step'], j['graph_update_leg'])"
