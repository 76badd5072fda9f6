#!/bin/bash
# counters of the default correspondence kernel (k_corr_flat) on the real bench configuration (4541 scans, 256 chains x lead 7)
TAG=${1:-flat}
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  LMONO_ODOM_STREAMS=${LMONO_ODOM_STREAMS:-1} timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-extras > $OUT/g$i.out 2> $OUT/g$i.err || { echo "group $i failed"; tail -3 $OUT/g$i.err; }
  python3 scripts/pmc_summary.py $OUT/g$i 2>&1 | grep -E "k_corr_flat|k_lm_solve|k_ring_sort" > $OUT/g$i.summary
  cat $OUT/g$i.summary
  find $OUT/g$i -name "*.csv" -size +1M -delete
done
