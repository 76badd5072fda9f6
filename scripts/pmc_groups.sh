#!/bin/bash
# Generic counter passes over the headline bench (one chain group so that the search launches do not overlap): one rocprofv3 --pmc run per
# counter group given on stdin (one group per line), per-launch averages of the kernels matching KERNELS.
#   usage: KERNELS="k_corr_flat|k_lm_solve" bash scripts/pmc_groups.sh <tag> ["<extra hipcc flags>"] [bench args] < groups.txt
TAG=${1:-pmc}; FLAGS=$2; shift; shift
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
if [ -n "$FLAGS" ]; then
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 $FLAGS -o $OUT/variant.so lmono_amd/csrc/lmono_hip.hip 2>$OUT/build.err || { tail -5 $OUT/build.err; exit 1; }
  export LMONO_HIP_LIB=$OUT/variant.so
fi
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  LMONO_ODOM_STREAMS=${STREAMS:-1} timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-extras "$@" > $OUT/g$i.out 2> $OUT/g$i.err || { echo "group $i failed"; tail -3 $OUT/g$i.err; }
  python3 scripts/pmc_summary.py $OUT/g$i 2>&1 | grep -E "${KERNELS:-k_corr_flat}" | tee -a $OUT/summary.txt
  rm -rf $OUT/g$i
done
rm -f $OUT/variant.so
