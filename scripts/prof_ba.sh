#!/bin/bash
# BA half under rocprofv3 (through gpurun): kernel-trace summary of the batched solve + MFMA / VALU counters of k_ba_solve.
# usage: bash scripts/prof_ba.sh <tag>
TAG=${1:-ba}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u > $OUT/mfma_counters.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --workload ba --windows 1024 --steps 3 --warmup 1 > $OUT/bench_ba.json 2> $OUT/trace.err
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
i=0
for grp in "SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_ANY" \
           "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc$i -- python3 bench.py --workload ba --windows 1024 --steps 1 --warmup 0 > $OUT/pmc$i.out 2> $OUT/pmc$i.err || { echo "pmc group $i failed"; tail -2 $OUT/pmc$i.err; }
  python3 scripts/pmc_summary.py $OUT/pmc$i 2>&1 | grep "k_ba_solve" > $OUT/pmc$i.summary
  cat $OUT/pmc$i.summary
done
find $OUT -name "*.csv" ! -name "kernel_stats.csv" -size +1M -delete
find $OUT -name "*.db" -delete
cat $OUT/mfma_counters.txt | tr "\n" " "; echo
head -8 $OUT/kernel_stats.csv
tail -c 700 $OUT/bench_ba.json
