#!/bin/bash
# headline bench over the number of chains / lead (after a kernel change moved the balance).  usage: bash scripts/chains_sweep.sh
mkdir -p gpurun_out/chains
for cfg in "192 4" "256 4" "320 4" "384 4" "256 3" "256 5" "320 5"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --no-extras --cpu-sample 0 --chains $1 --lead $2 > gpurun_out/chains/c$1l$2.json 2> gpurun_out/chains/c$1l$2.err || { tail -3 gpurun_out/chains/c$1l$2.err; continue; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/chains/c$1l$2.json").read().strip().splitlines()[-1])
print("chains $1 lead $2:", d["value"], d["ms_per_step"], "ate", d["ate_vs_cpu_m"], "odo", d["roofline"]["group_ms_per_step"]["odometry_total"], "repair", d["boundary_validation"]["repair_ms_per_step"], "rerun", d["boundary_validation"]["pairs_rerun"])
PY
done
