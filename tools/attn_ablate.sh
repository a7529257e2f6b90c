#!/bin/bash
# Where does the forward attention kernel spend its time?  Ablation builds of csrc/attention.hip (each computes garbage) timed on
# tools/attn_scale.py.  Build here (no GPU needed): tools/attn_ablate.sh build     Run on the GPU box: tools/attn_ablate.sh run [len] [S list]
R=${GRAFT_REPO_ROOT:-/root/repo}
C=$R/recurrent-offpolicy-rl_amd/csrc
B=$R/tools/micro/bin
VARS="${ATTN_VARS:-NOMAX NORESCALE NOEXP NOSUM}"
if [ "$1" = build ]; then
  mkdir -p $B/ab
  for v in $VARS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form -I$R/include $(for f in ${v//_/ }; do echo -n "-DATTN_AB_$f "; done) -c $C/attention.hip -o $B/ab/attn_$v.o &
  done
  wait
  for v in $VARS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $C/build/*.o | grep -v attention.o) $B/ab/attn_$v.o -o $B/libresel_attn_$v.so
  done
  ls -la $B/libresel_attn_*.so
else
  L=${2:-1026}; SL=${3:-32,128}
  echo "product:"; python3 $R/tools/attn_scale.py $L $SL
  for v in $VARS; do echo "ablation $v:"; RESEL_HIP_LIBRARY=$B/libresel_attn_$v.so python3 $R/tools/attn_scale.py $L $SL; done
fi
