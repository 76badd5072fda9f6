#!/bin/bash
# texture-addresser / L1 / TLB counters of the odometry kernels (one rocprofv3 --pmc pass per group; bench --scans 512 --steps 1)
TAG=${1:-ta}
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for grp in "TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE TA_FLAT_READ_WAVEFRONTS_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN2_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 bench.py --scans 512 --steps 1 --warmup 0 --cpu-sample 0 > $OUT/g$i.out 2> $OUT/g$i.err || { echo "group $i failed"; tail -3 $OUT/g$i.err; }
  python3 scripts/pmc_summary.py $OUT/g$i 2>&1 | grep "k_correspond\|k_lm_solve\|k_voxel\|k_select" > $OUT/g$i.summary
  cat $OUT/g$i.summary
  find $OUT/g$i -name "*.csv" -size +1M -delete
done
