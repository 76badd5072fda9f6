#!/bin/bash
# Kernel-time breakdown of laserMapping (scripts/bench_mapper.py, 64 frames, one stream) under rocprofv3.  usage: bash scripts/prof_mapper.sh
set -u
OUT=$PWD/gpurun_out/prof_mapper
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 scripts/bench_mapper.py 64 > $OUT/out.txt 2> $OUT/err.txt
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
rm -rf $OUT/trace
grep "k_map\|k_vox_\|k_grid_\|k_copy\|k_scatter\|rocclr" $OUT/kernel_stats.csv | cut -d, -f1-4 | cut -c1-140
