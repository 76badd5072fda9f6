#!/bin/bash
# The two front-end fusions of VERDICT r4 #2 as compile-time variants (scratch libraries on the GPU box): every LiDAR parity test (bit-exact curvature,
# selections, clouds, odometry), then the headline bench with the per-kernel device times.  usage: bash scripts/fe_fusion_sweep.sh <tag>
TAG=${1:-fe}; OUT=gpurun_out/$TAG; mkdir -p $OUT
i=0
while read -r flags; do
  i=$((i+1)); lib=$PWD/$OUT/v$i.so
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 $flags -o $lib lmono_amd/csrc/lmono_hip.hip 2>$OUT/build$i.err || { echo "build failed: $flags"; tail -3 $OUT/build$i.err; continue; }
  LMONO_HIP_LIB=$lib timeout -k 10 400 python -m pytest tests/test_lidar_gpu.py tests/test_sequence_gpu.py -m gpu -x -q > $OUT/t$i.log 2>&1; rc=$?
  LMONO_HIP_LIB=$lib timeout -k 10 250 python bench.py --no-extras --cpu-sample 0 > $OUT/b$i.json 2> $OUT/b$i.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/b$i.json").read().strip().splitlines()[-1]); g=d["roofline"]["group_ms_per_step"]
    print("[$flags] tests rc=$rc (%s) | %.0f scans/s %.3f ms | front end %.3f: sort %.3f curvature %.3f select %.3f voxel %.3f compact(+index) %.3f line index %.3f | odometry %.3f" % (open("$OUT/t$i.log").read().strip().splitlines()[-1], d["value"], d["ms_per_step"], g["frontend_total"], g["k_ring_sort"], g["k_curvature"], g["k_select"], g["k_voxel"], g["k_compact"], g["k_line_index"], g["odometry_total"]), flush=True)
except Exception as e:
    print("[$flags] tests rc=$rc | bench failed:", e, flush=True)
PY
  rm -f $lib
done | tee $OUT/sweep.txt
