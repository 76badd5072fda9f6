#!/bin/bash
# Kernel-trace summary of the front end only (scanreg over the headline batch, odometry with one chain group kept short).  usage: bash scripts/prof_frontend.sh <tag>
set -u
TAG=${1:-fe}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
LMONO_BOUNDARY_TOL=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-extras > $OUT/bench.json 2> $OUT/trace.err
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
rm -rf $OUT/trace
grep -v "corr\|lm_solve\|rocclr\|at::" $OUT/kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
