#!/bin/bash
# usage: tools/micro/build_lab.sh <output name> [extra -D flags...]   -> tools/micro/bin/<name> (+ <name>.s of the device code)
set -e
R=/root/repo
name=$1; shift
mkdir -p $R/tools/micro/bin /tmp/lab_$name
cd /tmp/lab_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Wno-unused-result "$@" $R/tools/micro/sscan_lab.hip -o $R/tools/micro/bin/$name -save-temps 2>&1 | grep -E "error" || true
cp /tmp/lab_$name/sscan_lab-hip-amdgcn-amd-amdhsa-gfx950.s $R/tools/micro/bin/$name.s
grep -E "\.name:|\.vgpr_count|vgpr_spill|\.sgpr_count" $R/tools/micro/bin/$name.s | paste - - - - | grep -E "sscan_(fwd|bwd)[0-9]*_kernelILi8ELi4" | sed 's/ \+/ /g'
