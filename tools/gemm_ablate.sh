#!/bin/bash
# Where does the bf16-split GEMM spend its time?  Three builds of the library (as shipped / no operand splitting / no matrix
# instructions; the last two compute garbage) timed on one shape.  Build here (no GPU needed): tools/gemm_ablate.sh build
# Run on the GPU box:  tools/gemm_ablate.sh run M N K
R=${GRAFT_REPO_ROOT:-/root/repo}
C=$R/recurrent-offpolicy-rl_amd/csrc
B=$R/tools/micro/bin
if [ "$1" = build ]; then
  mkdir -p $B/ab
  for v in NOSPLIT NOMFMA STAMP; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DGEMM_AB_$v -DGEMM_$v -c $C/gemm_f32.hip -o $B/ab/gemm_$v.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $C/build/*.o | grep -v gemm_f32.o) $B/ab/gemm_$v.o -o $B/libresel_$v.so
  done
  ls -la $B/*.so
else
  shift
  python3 $R/tools/prof_gemm.py "$@" 1 1 20 6
  for v in NOSPLIT NOMFMA; do echo "ablation $v:"; RESEL_HIP_LIBRARY=$B/libresel_$v.so python3 $R/tools/prof_gemm.py "$@" 1 1 20 6; done
  echo "phase stamps (cycles since the step began; block 0 wave 0):"
  RESEL_HIP_LIBRARY=$B/libresel_STAMP.so RESEL_GEMM_STAMPS=1 python3 $R/tools/prof_gemm.py "$@" 1 1 3 6
fi
