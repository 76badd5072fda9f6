#!/bin/bash
# address-translation counters of the search kernel (is the gather pattern bound by the per-CU TLB?), one chain group
#   usage: bash scripts/pmc_utcl.sh <tag> ["<extra hipcc flags>"] [bench args]
TAG=${1:-utcl}; FLAGS=$2; shift; shift
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
if [ -n "$FLAGS" ]; then
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 $FLAGS -o $OUT/variant.so lmono_amd/csrc/lmono_hip.hip 2>$OUT/build.err || { tail -5 $OUT/build.err; exit 1; }
  export LMONO_HIP_LIB=$OUT/variant.so
fi
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
i=0
for grp in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum GRBM_GUI_ACTIVE" \
           "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_LRU_INFLIGHT_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_MISSFIFO_FULL_sum GRBM_GUI_ACTIVE" \
           "TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  LMONO_ODOM_STREAMS=1 timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-extras "$@" > $OUT/g$i.out 2> $OUT/g$i.err || { echo "group $i failed"; tail -3 $OUT/g$i.err; }
  python3 scripts/pmc_summary.py $OUT/g$i 2>&1 | grep -E "k_corr_flat|k_ring_scatter|k_lm_solve" > $OUT/g$i.summary
  cat $OUT/g$i.summary
  rm -rf $OUT/g$i
done
rm -f $OUT/variant.so
