#!/bin/bash
# Round profile (run on the GPU box through gpurun): kernel-trace summary of the headline bench command, then the two
# HBM-traffic counters of k_correspond in their own passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass).
# usage: bash scripts/profile_round.sh <tag>
set -u
TAG=${1:-final}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --scans 1024 --steps 2 --cpu-sample 0 > $OUT/trace_bench.json 2> $OUT/trace.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --scans 512 --steps 1 --warmup 0 --cpu-sample 0 > $OUT/pmc_fetch.out 2> $OUT/pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --scans 512 --steps 1 --warmup 0 --cpu-sample 0 > $OUT/pmc_write.out 2> $OUT/pmc_write.err
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 bench.py --scans 512 --steps 1 --warmup 0 --cpu-sample 0 > $OUT/pmc_l2.out 2> $OUT/pmc_l2.err
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
for d in pmc_fetch pmc_write pmc_l2; do python3 scripts/pmc_summary.py $OUT/$d > $OUT/$d.summary 2>&1; done
# keep only the small summaries (gpurun_out is capped)
find $OUT -name "*.csv" ! -name "kernel_stats.csv" -size +2M -delete
ls -la $OUT
head -30 $OUT/kernel_stats.csv
cat $OUT/*.summary
tail -1 $OUT/trace_bench.json | cut -c1-400
