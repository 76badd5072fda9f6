#!/bin/bash
# chains x lead-in of the validated chained schedule (run-time arguments only): per-pass time, repair time, ATE
mkdir -p gpurun_out/chains_sweep
for cfg in "256 6 3" "320 6 3" "384 6 3" "512 6 3" "192 6 3" "384 5 3" "512 5 2" "320 7 3"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --no-extras --cpu-sample 0 --steps 10 --chains $1 --lead $2 --lead-full $3 > gpurun_out/chains_sweep/c$1_l$2_f$3.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/chains_sweep/c$1_l$2_f$3.json").read().strip().splitlines()[-1])
g=d["roofline"]["group_ms_per_step"]; v=d["boundary_validation"]
print("chains $1 lead $2 full $3 |", d["value"], d["ms_per_step"], "ate", d["ate_vs_cpu_m"], "odo", g["odometry_total"], "corr", g["k_correspond"], "lm", g["k_lm_solve"], "repair ms", v["repair_ms_per_step"], "flagged", v["flagged"], "pairs", v["pairs_rerun"], "ms/launch", d["roofline"]["ms_per_launch"], flush=True)
PY
done
