#!/bin/bash
# Judged profile artefacts of the bench command (run on the GPU box through gpurun):
#   1. rocprofv3 --kernel-trace --stats                       -> gpurun_out/prof_<tag>/<tag>_kernel_stats.csv
#   2. rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no trace domains) -> <tag>_fetch / <tag>_write
#   3. the plain bench line                                    -> gpurun_out/prof_<tag>/<tag>_bench.json
# usage: [PMC=0] profile_bench.sh <tag> [bench args...]        (PMC=0: kernel stats + bench line only)
TAG=${1:-r01}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
ARGS=${@:---steps 5 --warmup 2 --no-cpu-baseline --no-strict-leg --no-graph-update --no-rccl-leg --no-suite}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 $R/bench.py $ARGS 2>/dev/null | tail -1 > $OUT/${TAG}_bench.json
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o $TAG --output-format csv -- python3 $R/bench.py $ARGS > $OUT/${TAG}_trace.log 2>&1
if [ "${PMC:-1}" != "0" ]; then
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT -o ${TAG}_fetch --output-format csv -- python3 $R/bench.py $ARGS > $OUT/${TAG}_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT -o ${TAG}_write --output-format csv -- python3 $R/bench.py $ARGS > $OUT/${TAG}_write.log 2>&1
fi
rm -f $OUT/*_kernel_trace.csv          # tens of MB; the stats summary is what is kept
ls -la $OUT
