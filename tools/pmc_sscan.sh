#!/bin/bash
# PMC passes over the standalone selective-scan driver (run on the GPU box through gpurun; counters only, no trace domains).
# usage: pmc_sscan.sh [passes...]   (default: all four)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${PMC_OUT:-pmc_sscan}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SETS=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
      "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
      "FETCH_SIZE" "WRITE_SIZE")
PASSES=${@:-1 2 3 4}
for i in $PASSES; do
  timeout 300 rocprofv3 --pmc ${SETS[$((i-1))]} -d $OUT -o p$i --output-format csv -- python3 $R/tools/${PMC_DRIVER:-prof_sscan.py} > $OUT/p$i.log 2>&1
done
ls $OUT
