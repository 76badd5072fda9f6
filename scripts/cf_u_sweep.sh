#!/bin/bash
# k_corr_flat: gathers in flight per lane (kCfU) x runs per lane (kCfPer), built on the box: index-exact tests + headline bench per variant.
mkdir -p gpurun_out/cfu
cp lmono_amd/lib/liblmono_hip.so gpurun_out/cfu/keep.so
i=0
while read -r flags; do
  i=$((i+1))
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 $flags -o lmono_amd/lib/liblmono_hip.so lmono_amd/csrc/lmono_hip.hip 2>gpurun_out/cfu/build$i.err || { echo "build failed: $flags"; tail -3 gpurun_out/cfu/build$i.err; continue; }
  timeout -k 10 200 python -m pytest tests/test_lidar_gpu.py -m gpu -x -q -k "correspondences or odometry_sequential or odometry_full or dense_rings or other_sensors_and_near or chain_sharded" > gpurun_out/cfu/t$i.log 2>&1; rc=$?
  timeout -k 10 200 python bench.py --no-extras --cpu-sample 0 > gpurun_out/cfu/b$i.json 2> gpurun_out/cfu/b$i.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/cfu/b$i.json").read().strip().splitlines()[-1])
g=d["roofline"]["group_ms_per_step"]
print("[$flags] tests rc=$rc |", d["value"], d["ms_per_step"], "ate", d["ate_vs_cpu_m"], "corr", g["k_correspond"], "odo", g["odometry_total"], "ms/launch", d["roofline"]["ms_per_launch"])
PY
done <<'VAR'
-DLMONO_CF_PER=12
-DLMONO_CF_PER=14
-DLMONO_CF_PER=16
-DLMONO_CF_PER=12 -DLMONO_CF_U=8
VAR
cp gpurun_out/cfu/keep.so lmono_amd/lib/liblmono_hip.so; rm gpurun_out/cfu/keep.so
