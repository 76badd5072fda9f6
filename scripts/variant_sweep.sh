#!/bin/bash
# Compile-time variants of the library (one line of hipcc flags per variant on stdin), built on the GPU box into SCRATCH libraries under
# gpurun_out/<tag>/ and selected through LMONO_HIP_LIB -- the product library lmono_amd/lib/liblmono_hip.so is never touched.  Per variant:
# the index-exact LiDAR tests (unless TESTS=0), then the headline bench.
#   usage: bash scripts/variant_sweep.sh <tag> [bench args] < variants.txt
TAG=${1:-sweep}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
i=0
while read -r flags; do
  i=$((i+1))
  lib=$PWD/$OUT/v$i.so
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 $flags -o $lib lmono_amd/csrc/lmono_hip.hip 2>$OUT/build$i.err || { echo "build failed: $flags"; tail -3 $OUT/build$i.err; continue; }
  rc=skipped
  if [ "${TESTS:-1}" != "0" ]; then
    LMONO_HIP_LIB=$lib timeout -k 10 300 python -m pytest tests/test_lidar_gpu.py -m gpu -x -q -k "correspondences or odometry_sequential or odometry_full or dense_rings or other_sensors_and_near or chain_sharded" > $OUT/t$i.log 2>&1; rc=$?
  fi
  LMONO_HIP_LIB=$lib timeout -k 10 250 python bench.py --no-extras --cpu-sample 0 "$@" > $OUT/b$i.json 2> $OUT/b$i.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/b$i.json").read().strip().splitlines()[-1])
    g=d["roofline"]["group_ms_per_step"]
    print("[$flags] tests rc=$rc |", d["value"], d["ms_per_step"], "ate", d["ate_vs_cpu_m"], "corr", g["k_correspond"], "lm", g["k_lm_solve"], "odo", g["odometry_total"], "front", g["frontend_total"], "ms/launch", d["roofline"]["ms_per_launch"], flush=True)
except Exception as e:
    print("[$flags] tests rc=$rc | bench failed:", e, flush=True)
PY
  rm -f $lib
done
