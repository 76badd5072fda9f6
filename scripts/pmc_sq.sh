#!/bin/bash
# SQ instruction / wait counters of the hot kernels (bench --scans 512 --steps 1)
TAG=${1:-sq}
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 bench.py --scans 512 --steps 1 --warmup 0 --cpu-sample 0 --no-extras > $OUT/g$i.out 2> $OUT/g$i.err || { echo "group $i failed"; tail -3 $OUT/g$i.err; }
  python3 scripts/pmc_summary.py $OUT/g$i 2>&1 | grep "k_" > $OUT/g$i.summary
  cat $OUT/g$i.summary
  find $OUT/g$i -name "*.csv" -size +1M -delete
done
