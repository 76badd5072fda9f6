#!/bin/bash
# kernel-trace of ONE headline step with the per-dispatch timeline kept (start/end of every launch, all streams).
# usage: bash scripts/prof_timeline.sh <tag> [bench args]
TAG=${1:-tl}; shift
OUT=$PWD/gpurun_out/tl_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 1 --warmup 1 --cpu-sample 0 --no-extras "$@" > $OUT/bench.json 2> $OUT/trace.err
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_trace.csv
rm -rf $OUT/trace
ls -la $OUT
head -12 $OUT/kernel_stats.csv
