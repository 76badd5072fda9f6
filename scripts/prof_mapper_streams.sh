#!/bin/bash
# Kernel-time breakdown of laserMapping (bench.py --workload map, 64 scans) under rocprofv3 for 1 and 64 streams, plus the host-side phase clocks
# (LMONO_MAP_PROF) of a plain run.  usage (on the GPU box): bash scripts/prof_mapper_streams.sh <tag>
set -u
TAG=${1:-run}
OUT=$PWD/gpurun_out/prof_map_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
for S in 1 64; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace$S -- python3 bench.py --workload map --scans 64 --streams $S > $OUT/bench_${S}streams_under_rocprof.json 2> $OUT/err$S.txt || exit 1
  find $OUT/trace$S -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_${S}streams.csv
  rm -rf $OUT/trace$S
  LMONO_MAP_PROF=1 timeout -k 10 120 python3 bench.py --workload map --scans 64 --streams $S 2> $OUT/phases_raw$S.txt > /dev/null || exit 1
  grep MAPPROF $OUT/phases_raw$S.txt | tail -200 | awk '{for(i=4;i<=NF;i+=2){s[$(i)]+=$(i+1)}; n++} END{printf "mean ms per frame batch over %d calls:", n; for(k in s) printf " %s %.3f", k, s[k]/n; printf "\n"}' > $OUT/phases_${S}streams.txt
  rm -f $OUT/phases_raw$S.txt
done
cut -d, -f1-4 $OUT/kernel_stats_1streams.csv | cut -c1-120 | head -24
cut -d, -f1-4 $OUT/kernel_stats_64streams.csv | cut -c1-120 | head -24
cat $OUT/phases_1streams.txt $OUT/phases_64streams.txt
