#!/bin/bash
# kernel-trace summary of a short headline run (through gpurun).  usage: bash scripts/prof_trace.sh <tag> [bench args]
TAG=${1:-t}; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --scans 1024 --steps 2 --warmup 1 --cpu-sample 0 --no-extras "$@" > $OUT/bench.json 2> $OUT/trace.err
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT -name "*.csv" ! -name "kernel_stats.csv" -size +1M -delete
find $OUT -name "*.db" -delete
head -25 $OUT/kernel_stats.csv
tail -c 600 $OUT/bench.json
