#!/bin/bash
# Round-4 profile of the HEADLINE configuration (run on the GPU box through gpurun): rocprofv3 kernel-trace summary of the default bench
# command, the same with the boundary validation off (main-pass launches only: the per-launch average bench.py's own events measure), the HBM
# traffic counters in their own passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; never together with a trace), and
# kernel-trace summaries of the secondary workloads (ba-seq, map).   usage: bash scripts/profile_round4.sh <tag>
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
: > $OUT/pmc.txt
for grp in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  LMONO_BOUNDARY_TOL=0 timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$tag -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-extras > $OUT/pmc_$tag.out 2> $OUT/pmc_$tag.err
  echo "## $grp" >> $OUT/pmc.txt
  python3 scripts/pmc_summary.py $OUT/pmc_$tag >> $OUT/pmc.txt 2>&1
  rm -rf $OUT/pmc_$tag
done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_baseq -- python3 bench.py --workload ba-seq --frames-seq 600 --cpu-frames 0 > $OUT/bench_ba_seq_600frames.json 2> $OUT/trace_baseq.err
find $OUT/trace_baseq -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/ba_seq_kernel_stats_600frames.csv
rm -rf $OUT/trace_baseq
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_map -- python3 bench.py --workload map --scans 64 --streams 1 > $OUT/bench_map_1stream.json 2> $OUT/trace_map.err
find $OUT/trace_map -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/map_kernel_stats_1stream.csv
rm -rf $OUT/trace_map
head -8 $OUT/kernel_stats.csv | cut -c1-200
head -4 $OUT/kernel_stats_main_pass_only.csv | cut -c1-200
grep "corr_flat" $OUT/pmc.txt
head -5 $OUT/ba_seq_kernel_stats_600frames.csv | cut -c1-160
head -8 $OUT/map_kernel_stats_1stream.csv | cut -c1-160
tail -1 $OUT/trace_bench.json | cut -c1-200
