#!/bin/bash
# Where does the split-once GEMM (csrc/gemm_bf3.hip) spend its time?  Ablation builds of the library (each computes garbage) timed on one shape.
# Build here (no GPU needed): tools/bf3_ablate.sh build      Run on the GPU box: tools/bf3_ablate.sh run M N K [akc bkc]
# BF3_VARS: variants, '+' joins several switches in one build (NOMFMA+NOEPI); BF3_MODE: product mode (default 2)
R=${GRAFT_REPO_ROOT:-/root/repo}
C=$R/recurrent-offpolicy-rl_amd/csrc
B=$R/tools/micro/bin
VARS="${BF3_VARS:-NOSPLIT NOMFMA NOBAR NOEPI NOLOAD NOMFMA+NOEPI NOMFMA+NOEPI+NOSPLIT CLOCK}"
MODE=${BF3_MODE:-2}
if [ "$1" = build ]; then
  mkdir -p $B/ab
  for v in $VARS; do
    D=""; for q in ${v//+/ }; do D="$D -DBF3_AB_$q"; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize $D -c $C/gemm_bf3.hip -o $B/ab/bf3_$v.o &
  done
  wait
  for v in $VARS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $C/build/*.o | grep -v gemm_bf3.o) $B/ab/bf3_$v.o -o $B/libresel_bf3_$v.so
  done
  ls -la $B/libresel_bf3_*.so
else
  shift
  AK=${4:-1}; BKc=${5:-1}
  python3 $R/tools/prof_gemm.py $1 $2 $3 $AK $BKc 20 $MODE
  for v in $VARS; do
    [ $v = CLOCK ] && continue
    echo "ablation $v:"; RESEL_HIP_LIBRARY=$B/libresel_bf3_$v.so python3 $R/tools/prof_gemm.py $1 $2 $3 $AK $BKc 20 $MODE
  done
  if [ -f $B/libresel_bf3_CLOCK.so ]; then   # 2 s of back-to-back launches, then the clock of the last one
    RESEL_BF3_CLOCK=1 RESEL_HIP_LIBRARY=$B/libresel_bf3_CLOCK.so python3 $R/tools/prof_gemm.py $1 $2 $3 $AK $BKc 3000 $MODE
  fi
fi
