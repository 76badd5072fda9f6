#!/bin/bash
# memory-pipeline counters of the default correspondence kernel (one chain group, so that launches do not overlap)
#   usage: bash scripts/pmc_tcp.sh <tag> ["<extra hipcc flags>": the counters of a compile-time variant, built into a scratch library]
TAG=${1:-tcp}
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
if [ -n "$2" ]; then
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 $2 -o $OUT/variant.so lmono_amd/csrc/lmono_hip.hip 2>$OUT/build.err || { tail -5 $OUT/build.err; exit 1; }
  export LMONO_HIP_LIB=$OUT/variant.so
fi
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
i=0
for grp in "TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  LMONO_ODOM_STREAMS=1 timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-extras > $OUT/g$i.out 2> $OUT/g$i.err || { echo "group $i failed"; tail -3 $OUT/g$i.err; }
  python3 scripts/pmc_summary.py $OUT/g$i 2>&1 | grep -E "k_corr_flat|k_ring_tag|k_ring_scatter|k_select|k_voxel<9|k_lm_solve" > $OUT/g$i.summary
  cat $OUT/g$i.summary
  rm -rf $OUT/g$i
done
rm -f $OUT/variant.so
