timeout 900 python3 -m pytest tests/test_trainer_gpu.py tests/test_cgpt_dropout.py -m gpu -x -q -k "graphed or offset_base" 2>&1 | tail -3
for cfg in "" "--rnn cgpt_h8_l6_p0.1_ml1024_rms --algo td3 --rows 32 --horizon 1024" "--rnn gilr --algo sac --rows 16 --horizon 2000"; do
for i in 1 2; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-strict-leg --no-rccl-leg --no-suite $cfg 2>gpurun_out/err.txt | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg:', round(j['ms_per_step'],3), 'eager', round(j['eager_ms_per_step'],3))" || tail -5 gpurun_out/err.txt; done; done
python3 tools/soak.py smamba_s32_c16_b2_nln 300 graph 2>&1 | grep -v amdgpu.ids | tail -2
