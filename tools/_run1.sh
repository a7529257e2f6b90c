timeout 2000 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
C="--steps 5 --warmup 2 --no-cpu-baseline --no-strict-leg --no-graph-update --no-rccl-leg --no-suite"
bash tools/profile_bench.sh r04f > /dev/null 2>&1
bash tools/profile_bench.sh r04f_cgpt $C --rnn cgpt_h8_l6_p0.1_ml1024_rms --algo td3 --rows 32 --horizon 1024 > /dev/null 2>&1
bash tools/profile_bench.sh r04f_gilr $C --rnn gilr --algo sac --rows 16 --horizon 2000 > /dev/null 2>&1
bash tools/profile_bench.sh r04f_lru $C --rnn lru --algo sac --rows 16 --horizon 2000 > /dev/null 2>&1
bash tools/profile_bench.sh r04f_gru $C --rnn gru --algo sac --rows 64 --horizon 1024 > /dev/null 2>&1
ls gpurun_out/prof_r04f* | grep -c csv
