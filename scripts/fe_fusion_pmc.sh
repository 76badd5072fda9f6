#!/bin/bash
# HBM traffic of the front-end kernels in the three fusion variants of VERDICT r4 #2 (scratch libraries on the GPU box): FETCH_SIZE and WRITE_SIZE
# in separate rocprofv3 passes of one headline pass over the 4541 scans, per kernel, in KB per launch (FETCH_SIZE counts 64 B per 128-B request on
# gfx950: double it for bytes).  usage: bash scripts/fe_fusion_pmc.sh <tag>
TAG=${1:-fepmc}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
i=0
: > $OUT/pmc_table.txt
while read -r flags; do
  i=$((i+1)); lib=$OUT/v$i.so
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 $flags -o $lib lmono_amd/csrc/lmono_hip.hip 2>$OUT/build$i.err || { echo "build failed: $flags"; continue; }
  echo "### $flags" >> $OUT/pmc_table.txt
  for grp in FETCH_SIZE WRITE_SIZE; do
    LMONO_HIP_LIB=$lib LMONO_BOUNDARY_TOL=0 timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-extras > /dev/null 2> $OUT/pmc$i.err
    python3 scripts/pmc_summary.py $OUT/pmc 2>&1 | grep -E "k_curvature|k_select|k_compact|k_line_index|k_voxel<9" >> $OUT/pmc_table.txt
    rm -rf $OUT/pmc
  done
  rm -f $lib
done < scripts/variants_r5_frontend_fusions.txt
cat $OUT/pmc_table.txt
