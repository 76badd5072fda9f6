#!/bin/bash
# Round-3 profile of the HEADLINE configuration (run on the GPU box through gpurun): kernel-trace summary of the default bench command
# (main pass + the boundary validation's repair launches), the same with the validation off (LMONO_BOUNDARY_TOL=0: main-pass launches
# only -- the per-launch average bench.py's own events measure), then the HBM-traffic counters in their own passes
# (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; never together with a trace).  usage: bash scripts/profile_round3.sh <tag>
set -u
TAG=${1:-final}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-extras > $OUT/trace_bench.json 2> $OUT/trace.err
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
rm -rf $OUT/trace
LMONO_BOUNDARY_TOL=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace0 -- python3 bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-extras > $OUT/trace_bench_no_validation.json 2> $OUT/trace0.err
find $OUT/trace0 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_main_pass_only.csv
rm -rf $OUT/trace0
for grp in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  LMONO_BOUNDARY_TOL=0 timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$tag -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-extras > $OUT/pmc_$tag.out 2> $OUT/pmc_$tag.err
  echo "## $grp" >> $OUT/pmc.txt
  python3 scripts/pmc_summary.py $OUT/pmc_$tag >> $OUT/pmc.txt 2>&1
  rm -rf $OUT/pmc_$tag
done
head -8 $OUT/kernel_stats.csv | cut -c1-200
head -4 $OUT/kernel_stats_main_pass_only.csv | cut -c1-200
cat $OUT/pmc.txt | grep "corr_flat\|ring_sort"
tail -1 $OUT/trace_bench.json | cut -c1-200
