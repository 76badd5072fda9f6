#!/bin/bash
# Radius-ladder variants of the search (compile-time constants), built on the box: correctness (index-exact tests) + headline bench per variant.
mkdir -p gpurun_out/ladder
cp lmono_amd/lib/liblmono_hip.so gpurun_out/ladder/keep.so
i=0
while read -r flags; do
  i=$((i+1))
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 $flags -o lmono_amd/lib/liblmono_hip.so lmono_amd/csrc/lmono_hip.hip 2>/dev/null || { echo "build failed: $flags"; continue; }
  timeout -k 10 200 python -m pytest tests/test_lidar_gpu.py -m gpu -x -q -k "correspondences or odometry_sequential or odometry_full or dense_rings or other_sensors_and_near" > gpurun_out/ladder/t$i.log 2>&1; rc=$?
  timeout -k 10 200 python bench.py --no-extras --cpu-sample 0 > gpurun_out/ladder/b$i.json 2> gpurun_out/ladder/b$i.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/ladder/b$i.json").read().strip().splitlines()[-1])
g=d["roofline"]["group_ms_per_step"]
print("[$flags] tests rc=$rc |", d["value"], d["ms_per_step"], "ate", d["ate_vs_cpu_m"], "corr", g["k_correspond"], "odo", g["odometry_total"], "ms/launch", d["roofline"]["ms_per_launch"])
PY
done <<'VAR'
-DLMONO_WR_A0=0.3f -DLMONO_WR_B0=0.03f -DLMONO_WR_A1=1.0f -DLMONO_WR_B1=0.1f
-DLMONO_WR_A0=0.5f -DLMONO_WR_B0=0.05f -DLMONO_WR_A1=1.5f -DLMONO_WR_B1=0.15f
-DLMONO_WR_A0=0.4f -DLMONO_WR_B0=0.06f -DLMONO_WR_A1=2.0f -DLMONO_WR_B1=0.15f
-DLMONO_WR_A0=0.3f -DLMONO_WR_B0=0.03f -DLMONO_WR_A1=1.0f -DLMONO_WR_B1=0.1f -DLMONO_R0_PLANE=0.25f -DLMONO_R0_EDGE=0.7f
-DLMONO_WR_A0=0.3f -DLMONO_WR_B0=0.03f -DLMONO_WR_A1=1.0f -DLMONO_WR_B1=0.1f -DLMONO_R0_PLANE=0.1f -DLMONO_R0_EDGE=0.4f
-DLMONO_WR_A0=0.3f -DLMONO_WR_B0=0.03f -DLMONO_WR_A1=1.0f -DLMONO_WR_B1=0.1f -DLMONO_CF_U=8 -DLMONO_CF_PER=12
VAR
cp gpurun_out/ladder/keep.so lmono_amd/lib/liblmono_hip.so; rm gpurun_out/ladder/keep.so
