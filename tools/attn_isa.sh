#!/bin/bash
# Device ISA of attention.hip with the Makefile's flags -> /tmp/isa/attention.s; prints registers / scratch per kernel.
mkdir -p /tmp/isa
C=/root/repo/recurrent-offpolicy-rl_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form -I/root/repo/include -I$C -S --cuda-device-only -o /tmp/isa/attention.s $C/attention.hip 2>&1 | grep -v hip-link
grep -n "\.amdhsa_kernel \|; NumVgprs\|; ScratchSize\|; Occupancy" /tmp/isa/attention.s | paste - - - - | sed 's/.*amdhsa_kernel _ZN12_GLOBAL__N_1//' | awk '{print $1, $4, $7, $9}' | grep attn
