#!/bin/bash
# lead-in length x full-feature lead-in pairs of the chained schedule: per-pass time, repair time and repaired pairs (run-time arguments only)
mkdir -p gpurun_out/lead_sweep
for cfg in "4 2" "5 2" "6 2" "5 3" "6 3" "7 2" "3 2" "4 4"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --no-extras --cpu-sample 0 --steps 10 --lead $1 --lead-full $2 > gpurun_out/lead_sweep/l$1_f$2.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/lead_sweep/l$1_f$2.json").read().strip().splitlines()[-1])
g=d["roofline"]["group_ms_per_step"]; v=d["boundary_validation"]
print("lead $1 full $2 |", d["value"], d["ms_per_step"], "ate", d["ate_vs_cpu_m"], "odo", g["odometry_total"], "corr", g["k_correspond"], "lm", g["k_lm_solve"], "repair ms", v["repair_ms_per_step"], "flagged", v["flagged"], "pairs", v["pairs_rerun"], "q50/90/99", v["residual_q50_q90_q99"], flush=True)
PY
done
